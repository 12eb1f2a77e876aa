/* pyskani_amd._hitlist - the `Hit` objects of one query built in C.
 *
 * The reference's Hit is a PyO3 class filled from skani's AniEstResult (hit.rs:119-123, lib.rs:654-656). Here a query's psk_hit
 * records become `Hit` tuples (pyskani_amd/database.py) in one pass: from Python that pass costs ~0.3 us per hit under the
 * interpreter lock - more than the GPU work of a contig query, and what caps the rate of queries from several host threads
 * (lib.rs:569: the reference releases the GIL around the query so that threads scale). Pure host logic: no compute happens here.
 * database.py falls back to its own (identical) pure-Python construction when this module is not built. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include "../../include/pyskani_amd.h"

/* build(cls, records, n, query_name, names, keep) -> [cls(...)] * n
 * records: a buffer of n psk_hit; item i = (ani, query_name, af_query, names[ref_index], af_ref, learned != 0, keep, i) */
static PyObject* build(PyObject* self, PyObject* args) {
    PyObject *cls, *recs, *qname, *names, *keep;
    Py_ssize_t n;
    if (!PyArg_ParseTuple(args, "OOnOOO", &cls, &recs, &n, &qname, &names, &keep)) return NULL;
    if (!PyType_Check(cls) || !PyType_IsSubtype((PyTypeObject*)cls, &PyTuple_Type)) { PyErr_SetString(PyExc_TypeError, "cls must be a tuple subclass"); return NULL; }
    if (!PyList_Check(names)) { PyErr_SetString(PyExc_TypeError, "names must be a list"); return NULL; }
    Py_buffer view;
    if (PyObject_GetBuffer(recs, &view, PyBUF_SIMPLE) != 0) return NULL;
    if (n < 0 || (size_t)view.len < (size_t)n * sizeof(psk_hit)) { PyBuffer_Release(&view); PyErr_SetString(PyExc_ValueError, "record buffer shorter than n hits"); return NULL; }
    const psk_hit* h = (const psk_hit*)view.buf;
    PyTypeObject* tp = (PyTypeObject*)cls;
    const Py_ssize_t n_names = PyList_GET_SIZE(names);
    PyObject* out = PyList_New(n);
    if (!out) { PyBuffer_Release(&view); return NULL; }
    for (Py_ssize_t i = 0; i < n; i++) {
        if ((Py_ssize_t)h[i].ref_index >= n_names) { PyErr_SetString(PyExc_IndexError, "ref_index beyond the database's names"); goto fail; }
        PyObject* t = tp->tp_alloc(tp, 8);
        if (!t) goto fail;
        PyObject* ref = PyList_GET_ITEM(names, (Py_ssize_t)h[i].ref_index);
        PyObject *a = PyFloat_FromDouble((double)h[i].ani), *q = PyFloat_FromDouble((double)h[i].af_query), *r = PyFloat_FromDouble((double)h[i].af_ref),
                 *idx = PyLong_FromSsize_t(i), *learned = h[i].learned ? Py_True : Py_False;
        if (!a || !q || !r || !idx) { Py_XDECREF(a); Py_XDECREF(q); Py_XDECREF(r); Py_XDECREF(idx); Py_DECREF(t); goto fail; }
        Py_INCREF(qname); Py_INCREF(ref); Py_INCREF(learned); Py_INCREF(keep);
        PyTuple_SET_ITEM(t, 0, a); PyTuple_SET_ITEM(t, 1, qname); PyTuple_SET_ITEM(t, 2, q); PyTuple_SET_ITEM(t, 3, ref);
        PyTuple_SET_ITEM(t, 4, r); PyTuple_SET_ITEM(t, 5, learned); PyTuple_SET_ITEM(t, 6, keep); PyTuple_SET_ITEM(t, 7, idx);
        PyList_SET_ITEM(out, i, t);
    }
    PyBuffer_Release(&view);
    return out;
fail:
    PyBuffer_Release(&view);
    Py_DECREF(out);
    return NULL;
}

static PyMethodDef methods[] = {{"build", build, METH_VARARGS, "Hit tuples of one query's psk_hit records"}, {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_hitlist", "C-level construction of pyskani_amd.Hit lists", -1, methods};
PyMODINIT_FUNC PyInit__hitlist(void) { return PyModule_Create(&moddef); }
