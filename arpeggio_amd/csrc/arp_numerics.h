// arp_numerics.h — device arithmetic of the hot path, written so that every
// rounding step matches the reference's NumPy expressions (DESIGN.md "Arithmetic
// model").  This translation unit is compiled with -ffp-contract=off: the only
// fused operations are the explicit fma() calls.
//
//   np.dot / np.linalg.norm on float32[3]: float32 products, float64 accumulation,
//       one rounding to float32 (OpenBLAS sdot tail loop), float32 sqrt.
//   np.dot / np.linalg.norm on float64[3]: fma(z,z',fma(y,y',x*x')), float64 sqrt.
//   NumPy scalar expressions (utils.py:712-745): one rounding per operation.
//   float32 (op) Python float: the Python float is cast to float32 first (NEP 50).
#pragma once
#include <hip/hip_runtime.h>

#define ARP_PI 3.141592653589793  // np.pi

namespace num {

struct d3 { double x, y, z; };
struct f3 { float x, y, z; };

__device__ __forceinline__ d3 to_d3(f3 v) { return {(double)v.x, (double)v.y, (double)v.z}; }
__device__ __forceinline__ d3 sub(d3 a, d3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 sub(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }

// np.dot(a, b), float32[3]
__device__ __forceinline__ float dot(f3 a, f3 b) {
    float p0 = a.x * b.x, p1 = a.y * b.y, p2 = a.z * b.z;
    double acc = 0.0;
    acc += (double)p0;
    acc += (double)p1;
    acc += (double)p2;
    return (float)acc;
}
// np.dot(a, b), float64[3]
__device__ __forceinline__ double dot(d3 a, d3 b) {
    double acc = a.x * b.x;
    acc = fma(a.y, b.y, acc);
    acc = fma(a.z, b.z, acc);
    return acc;
}
__device__ __forceinline__ float norm(f3 v) { return sqrtf(dot(v, v)); }
__device__ __forceinline__ double norm(d3 v) { return sqrt(dot(v, v)); }

// Bio.PDB.kdtrees membership test: float64 sum of squares, no fusion.
__device__ __forceinline__ double dist2_kd(d3 a, d3 b) {
    double dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
    double r = dx * dx;
    r += dy * dy;
    r += dz * dz;
    return r;
}

// utils.get_angle (utils.py:696-745), all-float64 operands
__device__ __forceinline__ double get_angle(d3 a, d3 b, d3 c) {
    d3 v1 = sub(a, b), v2 = sub(c, b);
    double m1 = sqrt(v1.x * v1.x + v1.y * v1.y + v1.z * v1.z);
    d3 n1 = {v1.x / m1, v1.y / m1, v1.z / m1};
    double m2 = sqrt(v2.x * v2.x + v2.y * v2.y + v2.z * v2.z);
    d3 n2 = {v2.x / m2, v2.y / m2, v2.z / m2};
    double res = n1.x * n2.x + n1.y * n2.y + n1.z * n2.z;
    double ang = acos(res);
    if (isnan(ang)) ang = ARP_PI;  // utils.py:741-743
    return ang;
}

// utils.get_angle with three float32 points (is_xbond, utils.py:174).
// nan_pi is set when the NaN -> np.pi substitution fired (np.pi is a Python float).
__device__ __forceinline__ float get_angle(f3 a, f3 b, f3 c, bool& nan_pi) {
    f3 v1 = sub(a, b), v2 = sub(c, b);
    float m1 = sqrtf(v1.x * v1.x + v1.y * v1.y + v1.z * v1.z);
    f3 n1 = {v1.x / m1, v1.y / m1, v1.z / m1};
    float m2 = sqrtf(v2.x * v2.x + v2.y * v2.y + v2.z * v2.z);
    f3 n2 = {v2.x / m2, v2.y / m2, v2.z / m2};
    float res = n1.x * n2.x + n1.y * n2.y + n1.z * n2.z;
    float ang = acosf(res);
    nan_pi = isnan(ang);
    return ang;
}

// utils.get_angle(nbr f32, halogen f32, hydrogen f64) (utils.py:151):
// v1 stays float32, v2 and the dot product are float64.
__device__ __forceinline__ double get_angle_mixed(f3 a, f3 b, d3 c) {
    f3 v1 = sub(a, b);
    d3 v2 = sub(c, to_d3(b));
    float m1 = sqrtf(v1.x * v1.x + v1.y * v1.y + v1.z * v1.z);
    f3 n1 = {v1.x / m1, v1.y / m1, v1.z / m1};
    double m2 = sqrt(v2.x * v2.x + v2.y * v2.y + v2.z * v2.z);
    d3 n2 = {v2.x / m2, v2.y / m2, v2.z / m2};
    double res = (double)n1.x * n2.x + (double)n1.y * n2.y + (double)n1.z * n2.z;
    double ang = acos(res);
    if (isnan(ang)) ang = ARP_PI;
    return ang;
}

// ---- decision shortcuts ---------------------------------------------------------------------------
// The reference decides "angle >= a_min" via normalise / dot / acos (2 sqrt, 6 divisions and an acos
// per hydrogen).  cosA = (v1.v2) / sqrt(|v1|^2 |v2|^2) is the same quantity up to ~1e-15 (1e-7 when one
// vector went through float32, utils.py:151), so away from the threshold by a margin delta the decision
// can be taken on squared quantities alone; the exact sequence above runs only inside the margin, when a
// vector has zero length (NaN -> pi, utils.py:741-743) or when |cosA| is within 1e-12 of 1 (rounding can
// push the reference's cosine past 1, which also ends in the NaN -> pi substitution).
__device__ __forceinline__ bool cos_le(double dot, double q, double t) {  // cosA <= t ?
    return (t >= 0) ? (dot <= 0 || dot * dot <= t * t * q) : (dot <= 0 && dot * dot >= t * t * q);
}
__device__ __forceinline__ bool cos_ge(double dot, double q, double t) {  // cosA >= t ?
    return (t >= 0) ? (dot >= 0 && dot * dot >= t * t * q) : (dot >= 0 || dot * dot <= t * t * q);
}
// 1: angle(a,b,c) >= a_min surely; 0: surely not; -1: run the exact test.  c_min = cos(a_min).
__device__ __forceinline__ int angle_ge_fast(d3 a, d3 b, d3 c, double c_min, double delta) {
    const d3 v1 = sub(a, b), v2 = sub(c, b);
    const double dot_ = v1.x * v2.x + v1.y * v2.y + v1.z * v2.z;
    const double q = (v1.x * v1.x + v1.y * v1.y + v1.z * v1.z) * (v2.x * v2.x + v2.y * v2.y + v2.z * v2.z);
    if (!(q > 0.0) || !(dot_ * dot_ < (1.0 - 2e-12) * q)) return -1;
    if (cos_le(dot_, q, c_min - delta)) return 1;
    if (cos_ge(dot_, q, c_min + delta)) return 0;
    return -1;
}
// 1: a_lo <= angle(a,b,c) <= a_hi surely; 0: surely not; -1: exact.  c_lo = cos(a_lo) > c_hi = cos(a_hi).
__device__ __forceinline__ int angle_in_fast(d3 a, d3 b, d3 c, double c_lo, double c_hi, double delta) {
    const d3 v1 = sub(a, b), v2 = sub(c, b);
    const double dot_ = v1.x * v2.x + v1.y * v2.y + v1.z * v2.z;
    const double q = (v1.x * v1.x + v1.y * v1.y + v1.z * v1.z) * (v2.x * v2.x + v2.y * v2.y + v2.z * v2.z);
    if (!(q > 0.0) || !(dot_ * dot_ < (1.0 - 2e-12) * q)) return -1;
    if (cos_le(dot_, q, c_lo - delta) && cos_ge(dot_, q, c_hi + delta)) return 1;
    if (cos_ge(dot_, q, c_lo + delta) || cos_le(dot_, q, c_hi - delta)) return 0;
    return -1;
}
// 1: sqrt(s) <= thr surely (s = exact sum of squares, thr2 = thr * thr); 0: surely not; -1: take the sqrt
__device__ __forceinline__ int dist_le_fast(double s, double thr2) {
    if (s <= thr2 * (1.0 - 1e-14)) return 1;
    if (s >= thr2 * (1.0 + 1e-14)) return 0;
    return -1;
}
#define ARP_COS_1_57 0.0007963267107332633
#define ARP_COS_2_27 (-0.6436084187135406)
#define ARP_COS_0_52 0.8678191796776499
#define ARP_COS_2_62 (-0.8670267214458024)

// degrees + "signed" folding, utils.py:656-660 / 689-693, then abs() at the call site
__device__ __forceinline__ double fold_deg(double rad) {
    if (rad > ARP_PI / 2) rad = rad - ARP_PI;
    return fabs(rad * 180 / ARP_PI);
}
__device__ __forceinline__ float fold_deg(float rad) {
    if (rad > (float)(ARP_PI / 2)) rad = rad - (float)ARP_PI;
    float t = rad * 180.0f;
    return fabsf(t / (float)ARP_PI);
}

// abs(group_angle(group, point, True, True)), utils.py:638-660
__device__ __forceinline__ double group_angle(d3 normal, d3 point) {
    double c = dot(normal, point) / (norm(normal) * norm(point));
    return fold_deg(acos(c));
}
__device__ __forceinline__ float group_angle(f3 normal, f3 point) {
    float c = dot(normal, point) / (norm(normal) * norm(point));
    return fold_deg(acosf(c));
}
// float32 normal against a float64 vector: the dot promotes, norm(normal) stays float32
__device__ __forceinline__ double group_angle(f3 normal, d3 other) {
    double c = dot(to_d3(normal), other) / ((double)norm(normal) * norm(other));
    return fold_deg(acos(c));
}

// interactions.py:1127-1148 (9 = '' when an angle is NaN)
__device__ __forceinline__ int pp_class(double dihedral, double theta) {
    if (dihedral <= 30.0 && theta <= 30.0) return 0;
    else if (dihedral <= 30.0 && theta <= 60.0) return 1;
    else if (dihedral <= 30.0 && theta <= 90.0) return 2;
    else if (30.0 < dihedral && dihedral <= 60.0 && theta <= 30.0) return 3;
    else if (30.0 < dihedral && dihedral <= 60.0 && theta <= 60.0) return 4;
    else if (30.0 < dihedral && dihedral <= 60.0 && theta <= 90.0) return 5;
    else if (60.0 < dihedral && dihedral <= 90.0 && theta <= 30.0) return 6;
    else if (60.0 < dihedral && dihedral <= 90.0 && theta <= 60.0) return 7;
    else if (60.0 < dihedral && dihedral <= 90.0 && theta <= 90.0) return 8;
    return 9;
}

}  // namespace num
