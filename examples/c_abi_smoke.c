/* The C ABI of libarpeggio_hip.so from plain C (no Python, no C++): packs a small made-up structure into a blob with the
 * host-only packer, reads a category of an mmCIF text with the native tokenizer, and — when a GPU is there — uploads the
 * blob, runs the whole run_arpeggio path and fetches the contacts.
 *
 *   gcc -std=c99 -I include examples/c_abi_smoke.c -o c_abi_smoke -L arpeggio_amd/csrc -larpeggio_hip -Wl,-rpath,$PWD/arpeggio_amd/csrc -lm
 *
 * Exit code 0 = everything that could run here ran and agreed with itself (prints "NO_GPU" or "GPU_OK"). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "arpeggio_hip.h"

#define N 600

static unsigned long long rng_state = 88172645463325252ull;
static double rnd(void) {   /* xorshift64 */
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) / 9007199254740992.0;
}

int main(void) {
    /* ---- a structure: N atoms in a 20 A cube, 4 atoms per residue, carbon / oxygen radii, a few donors and acceptors */
    static float xyz[3 * N];
    static double vdw[N], cov[N];
    static uint16_t tmask[N], flags[N];
    static int32_t res_id[N], bond_off[N + 1], h_off[N + 1], sb_nbr[N];
    enum { NRES = N / 4 };
    static uint8_t res_flags[NRES];
    static int32_t res_prev[NRES], res_next[NRES];
    int i;
    for (i = 0; i < N; ++i) {
        xyz[3 * i] = (float)(20.0 * rnd()); xyz[3 * i + 1] = (float)(20.0 * rnd()); xyz[3 * i + 2] = (float)(20.0 * rnd());
        vdw[i] = (i % 3) ? 1.7 : 1.52; cov[i] = (i % 3) ? 0.76 : 0.66;
        tmask[i] = (uint16_t)((i % 5 == 0) ? ARP_T_HBOND_ACCEPTOR : ((i % 7 == 0) ? ARP_T_HYDROPHOBE : 0));
        flags[i] = 0; res_id[i] = i / 4; sb_nbr[i] = -1;
    }
    for (i = 0; i <= N; ++i) { bond_off[i] = 0; h_off[i] = 0; }
    for (i = 0; i < NRES; ++i) { res_flags[i] = 0; res_prev[i] = -1; res_next[i] = -1; }

    /* ---- host-only: blob */
    uint64_t bytes = arp_blob_size(N, NRES, 0, 0, 0, 0);
    void* blob = malloc(bytes);
    if (!bytes || !blob) return 10;
    if (arp_blob_layout(blob, bytes, N, NRES, 0, 0, 0, 0) != ARP_OK) return 11;
    if (arp_blob_fill(blob, bytes, xyz, vdw, cov, tmask, flags, res_id, res_flags, res_prev, res_next, bond_off, NULL, h_off, NULL, sb_nbr,
                      NULL, NULL, NULL, NULL, NULL, NULL) != ARP_OK) return 12;
    arp_blob_header hdr;
    memcpy(&hdr, blob, sizeof hdr);
    if (hdr.magic != ARP_BLOB_MAGIC || hdr.n != N || hdr.n_rad != 2) return 13;

    /* ---- host-only: mmCIF category */
    const char* cif = "data_X\nloop_\n_atom_site.id\n_atom_site.label_atom_id\n_atom_site.Cartn_x\n1 \"O5'\" 1.5\n2 CA ?\n";
    arp_cif* t = NULL;
    char err[128];
    if (arp_cif_open(cif, strlen(cif), "_atom_site.", &t, err, sizeof err) != ARP_OK) { fprintf(stderr, "%s\n", err); return 14; }
    double x[2];
    int64_t bad = -1;
    if (arp_cif_rows(t) != 2 || arp_cif_cols(t) != 3 || strcmp(arp_cif_tag(t, 1), "label_atom_id") != 0) return 15;
    if (arp_cif_column_f64(t, 2, -1.0, x, &bad) != ARP_OK || x[0] != 1.5 || x[1] != -1.0) return 16;
    arp_cif_close(t);

    /* ---- the device part */
    arp_ctx* ctx = NULL;
    if (arp_create(0, &ctx) != ARP_OK) {
        printf("NO_GPU (%s)\n", arp_last_error(NULL));
        free(blob);
        return 0;
    }
    if (arp_set_blob(ctx, blob, bytes) != ARP_OK) { fprintf(stderr, "%s\n", arp_last_error(ctx)); return 20; }
    int64_t counts[5];
    if (arp_run_launch(ctx, 5.0, 0.1, 0, 6.0, counts) != ARP_OK) { fprintf(stderr, "%s\n", arp_last_error(ctx)); return 21; }
    int64_t n = counts[0], got = 0;
    int32_t* ci = malloc(sizeof(int32_t) * (size_t)(n + 1)); int32_t* cj = malloc(sizeof(int32_t) * (size_t)(n + 1));
    float* cd = malloc(sizeof(float) * (size_t)(n + 1)); uint16_t* cs = malloc(sizeof(uint16_t) * (size_t)(n + 1));
    uint8_t* ct = malloc((size_t)(n + 1));
    if (arp_atom_contacts_fetch(ctx, n, ci, cj, cd, cs, ct, &got) != ARP_OK || got != n) return 22;
    /* every reported pair is a pair of different residues within the cutoff, and none is missing (brute force) */
    int64_t expect = 0, k;
    int a, b;
    for (a = 0; a < N; ++a)
        for (b = a + 1; b < N; ++b) {
            double dx = (double)xyz[3 * a] - xyz[3 * b], dy = (double)xyz[3 * a + 1] - xyz[3 * b + 1], dz = (double)xyz[3 * a + 2] - xyz[3 * b + 2];
            if (dx * dx + dy * dy + dz * dz <= 25.0 && res_id[a] != res_id[b]) ++expect;
        }
    for (k = 0; k < n; ++k)
        if (ci[k] >= cj[k] || res_id[ci[k]] == res_id[cj[k]] || !(cd[k] <= 5.0f) || (cs[k] & 0x1F) == 0) return 23;
    if (expect != n) { fprintf(stderr, "expected %lld contacts, got %lld\n", (long long)expect, (long long)n); return 24; }
    /* the canonical (bgn, end) order, made on the device, and every bag with ONE copy (arp_atom_contacts_sort, arp_fetch_packed) */
    if (arp_atom_contacts_sort(ctx) != ARP_OK) return 25;
    if (arp_atom_contacts_fetch(ctx, n, ci, cj, cd, cs, ct, &got) != ARP_OK || got != n) return 26;
    for (k = 1; k < n; ++k)
        if (ci[k - 1] > ci[k] || (ci[k - 1] == ci[k] && cj[k - 1] >= cj[k])) return 27;
    {
        uint64_t off[ARP_PACKED_OFFSETS], used = 0;
        int64_t pc[5];
        void* host = NULL;
        if (arp_fetch_packed(ctx, NULL, 0, pc, off, &used) != ARP_E_CAPACITY || used == 0) return 28;   /* asks for the size first */
        if (arp_host_alloc(used, &host) != ARP_OK) return 29;
        if (arp_fetch_packed(ctx, host, used, pc, off, &used) != ARP_OK || pc[0] != n) return 30;
        const int32_t* pi = (const int32_t*)((const char*)host + off[0]);
        const int32_t* pj = (const int32_t*)((const char*)host + off[1]);
        const float* pd = (const float*)((const char*)host + off[2]);
        for (k = 0; k < n; ++k)
            if (pi[k] != ci[k] || pj[k] != cj[k] || pd[k] != cd[k]) return 31;
        arp_host_free(host);
    }
    printf("GPU_OK %lld contacts\n", (long long)n);
    arp_destroy(ctx);
    free(ci); free(cj); free(cd); free(cs); free(ct); free(blob);
    return 0;
}
