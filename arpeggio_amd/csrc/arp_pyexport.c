/* arp_pyexport.c — CPython helper of core/export.py: the atom-atom part of get_contacts() (interactions.py:172-212) built
 * in C.  A whole-structure run has a million records; a Python-level dict comprehension spends ~3 us on each, this ~0.7 us.
 * Pure host code (no GPU, no HIP); export.py falls back to the comprehension when the module is not built.
 *
 *   gcc -O2 -shared -fPIC $(python3-config --includes) arp_pyexport.c -o ../_pyexport.so
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject* atom_atom_records(PyObject* self, PyObject* args) {
    (void)self;
    Py_buffer bi, bj, bd, bs, bc;
    PyObject *atom_dicts, *names_by_sift, *ctype_names;
    int share = 0;   /* 1: the records of an atom share ONE inner dictionary (and the records of a fingerprint one name list) */
    if (!PyArg_ParseTuple(args, "y*y*y*y*y*O!O!O!|p", &bi, &bj, &bd, &bs, &bc, &PyList_Type, &atom_dicts, &PyList_Type, &names_by_sift,
                          &PyList_Type, &ctype_names, &share))
        return NULL;
    PyObject* out = NULL;
    const Py_ssize_t n = bi.len / 4;
    if (bj.len / 4 != n || bd.len / 8 != n || bs.len / 2 != n || bc.len != n) {
        PyErr_SetString(PyExc_ValueError, "atom_atom_records: array lengths differ");
        goto done;
    }
    const int32_t* ci = (const int32_t*)bi.buf;
    const int32_t* cj = (const int32_t*)bj.buf;
    const double* cd = (const double*)bd.buf;
    const uint16_t* cs = (const uint16_t*)bs.buf;
    const uint8_t* cc = (const uint8_t*)bc.buf;
    const Py_ssize_t n_atoms = PyList_GET_SIZE(atom_dicts), n_sift = PyList_GET_SIZE(names_by_sift), n_ct = PyList_GET_SIZE(ctype_names);
    PyObject* k_bgn = PyUnicode_InternFromString("bgn");
    PyObject* k_end = PyUnicode_InternFromString("end");
    PyObject* k_type = PyUnicode_InternFromString("type");
    PyObject* k_dist = PyUnicode_InternFromString("distance");
    PyObject* k_contact = PyUnicode_InternFromString("contact");
    PyObject* k_ent = PyUnicode_InternFromString("interacting_entities");
    PyObject* v_type = PyUnicode_InternFromString("atom-atom");
    /* every record is a copy of one template (keys in place, one allocation for the table) whose values are then replaced */
    PyObject* tmpl = PyDict_New();
    if (!tmpl || PyDict_SetItem(tmpl, k_bgn, Py_None) || PyDict_SetItem(tmpl, k_end, Py_None) || PyDict_SetItem(tmpl, k_type, v_type) ||
        PyDict_SetItem(tmpl, k_dist, Py_None) || PyDict_SetItem(tmpl, k_contact, Py_None) || PyDict_SetItem(tmpl, k_ent, Py_None)) {
        Py_XDECREF(tmpl);
        goto keys;
    }
    out = PyList_New(n);
    if (!out) { Py_DECREF(tmpl); goto keys; }
    for (Py_ssize_t r = 0; r < n; ++r) {
        if (ci[r] < 0 || ci[r] >= n_atoms || cj[r] < 0 || cj[r] >= n_atoms || cs[r] >= n_sift || cc[r] >= n_ct) {
            PyErr_SetString(PyExc_IndexError, "atom_atom_records: index out of range");
            Py_CLEAR(out);
            break;
        }
        PyObject* ab = PyList_GET_ITEM(atom_dicts, ci[r]);
        PyObject* ae = PyList_GET_ITEM(atom_dicts, cj[r]);
        PyObject* nm = PyList_GET_ITEM(names_by_sift, cs[r]);
        if (!PyDict_Check(ab) || !PyDict_Check(ae) || !PyList_Check(nm)) {
            PyErr_SetString(PyExc_TypeError, "atom_atom_records: missing atom dictionary or name list");
            Py_CLEAR(out);
            break;
        }
        PyObject* rec = PyDict_Copy(tmpl);
        PyObject *b, *e, *names;
        if (share) { b = ab; e = ae; names = nm; Py_INCREF(b); Py_INCREF(e); Py_INCREF(names); }
        else { b = PyDict_Copy(ab); e = PyDict_Copy(ae); names = PyList_GetSlice(nm, 0, PyList_GET_SIZE(nm)); }
        PyObject* d = PyFloat_FromDouble(cd[r]);
        int bad = !rec || !b || !e || !d || !names;
        if (!bad) {
            bad = PyDict_SetItem(rec, k_bgn, b) || PyDict_SetItem(rec, k_end, e) ||
                  PyDict_SetItem(rec, k_dist, d) || PyDict_SetItem(rec, k_contact, names) ||
                  PyDict_SetItem(rec, k_ent, PyList_GET_ITEM(ctype_names, cc[r]));
        }
        Py_XDECREF(b); Py_XDECREF(e); Py_XDECREF(d); Py_XDECREF(names);
        if (bad) {
            Py_XDECREF(rec);
            Py_CLEAR(out);
            break;
        }
        PyList_SET_ITEM(out, r, rec);
    }
    Py_DECREF(tmpl);
keys:
    Py_XDECREF(k_bgn); Py_XDECREF(k_end); Py_XDECREF(k_type); Py_XDECREF(k_dist); Py_XDECREF(k_contact); Py_XDECREF(k_ent); Py_XDECREF(v_type);
done:
    PyBuffer_Release(&bi); PyBuffer_Release(&bj); PyBuffer_Release(&bd); PyBuffer_Release(&bs); PyBuffer_Release(&bc);
    return out;
}

static PyMethodDef methods[] = {
    {"atom_atom_records", atom_atom_records, METH_VARARGS,
     "atom_atom_records(i, j, distance, sift, ctype, atom_dicts, names_by_sift, ctype_names, share=False) -> list of record dicts"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_pyexport", "get_contacts() records in C", -1, methods, NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__pyexport(void) { return PyModule_Create(&module); }
