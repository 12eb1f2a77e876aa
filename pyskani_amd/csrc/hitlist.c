/* pyskani_amd._hitlist - the `Hit` objects of one query built in C.
 *
 * The reference's Hit is a PyO3 class filled from skani's AniEstResult (hit.rs:119-123, lib.rs:654-656). Here a query's psk_hit
 * records become `Hit` objects (pyskani_amd/database.py) in one pass: from Python that pass costs ~0.3 us per hit under the
 * interpreter lock - more than the GPU work of a contig query, and what caps the rate of queries from several host threads
 * (lib.rs:569: the reference releases the GIL around the query so that threads scale). `query_host` is the whole per-contig call
 * (arguments, the library call with the lock released, the Hit list) for the same reason. Pure host logic: no compute happens here.
 * database.py falls back to its own (identical) pure-Python construction when this module is not built. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include "../../include/pyskani_amd.h"

/* HitBase: the storage behind pyskani_amd.Hit - the five fields of the reference's class (hit.rs:77-104: getters only), `learned`, and the way back to the
 * library's record (`_keep`: the records of the query, `_idx`: which one). ONE allocation per hit: a hit made of a tuple, three floats and an int cost five,
 * and a hundred hits per contig query under the interpreter lock were what kept eight querying threads from scaling. */
typedef struct {
    PyObject_HEAD
    float ani, af_query, af_ref;
    uint32_t idx;
    PyObject *qname, *rname, *keep;      /* str, str, object (None for a hit built by hand) */
    unsigned char learned;
} HitObject;

static void hit_dealloc(HitObject* h) {
    Py_XDECREF(h->qname); Py_XDECREF(h->rname); Py_XDECREF(h->keep);
    Py_TYPE(h)->tp_free((PyObject*)h);
}
/* HitBase(identity, query_name, query_fraction, reference_name, reference_fraction, learned=False, keep=None, idx=0); values are stored as given
 * (float32): the range checks of hit.rs:34-48 live in pyskani_amd.Hit.__new__ */
static PyObject* hit_new(PyTypeObject* tp, PyObject* args, PyObject* kw) {
    static char* names[] = {"identity", "query_name", "query_fraction", "reference_name", "reference_fraction", "learned", "keep", "idx", NULL};
    double ani, afq, afr; PyObject *qn, *rn, *keep = Py_None; int learned = 0; unsigned int idx = 0;
    if (!PyArg_ParseTupleAndKeywords(args, kw, "dUdUd|pOI", names, &ani, &qn, &afq, &rn, &afr, &learned, &keep, &idx)) return NULL;
    HitObject* h = (HitObject*)tp->tp_alloc(tp, 0);
    if (!h) return NULL;
    h->ani = (float)ani; h->af_query = (float)afq; h->af_ref = (float)afr; h->idx = idx; h->learned = learned ? 1 : 0;
    Py_INCREF(qn); Py_INCREF(rn); Py_INCREF(keep);
    h->qname = qn; h->rname = rn; h->keep = keep;
    return (PyObject*)h;
}
static PyObject* hit_get_identity(HitObject* h, void* c) { return PyFloat_FromDouble((double)h->ani); }
static PyObject* hit_get_qfrac(HitObject* h, void* c) { return PyFloat_FromDouble((double)h->af_query); }
static PyObject* hit_get_rfrac(HitObject* h, void* c) { return PyFloat_FromDouble((double)h->af_ref); }
static PyObject* hit_get_qname(HitObject* h, void* c) { Py_INCREF(h->qname); return h->qname; }
static PyObject* hit_get_rname(HitObject* h, void* c) { Py_INCREF(h->rname); return h->rname; }
static PyObject* hit_get_keep(HitObject* h, void* c) { Py_INCREF(h->keep); return h->keep; }
static PyObject* hit_get_idx(HitObject* h, void* c) { return PyLong_FromUnsignedLong(h->idx); }
static PyObject* hit_get_learned(HitObject* h, void* c) { return PyBool_FromLong(h->learned); }
static PyGetSetDef hit_getset[] = {
    {"identity", (getter)hit_get_identity, NULL, "hit.rs:77-80", NULL}, {"query_name", (getter)hit_get_qname, NULL, "hit.rs:83-86", NULL},
    {"query_fraction", (getter)hit_get_qfrac, NULL, "hit.rs:89-92", NULL}, {"reference_name", (getter)hit_get_rname, NULL, "hit.rs:95-98", NULL},
    {"reference_fraction", (getter)hit_get_rfrac, NULL, "hit.rs:101-104", NULL},
    {"learned", (getter)hit_get_learned, NULL, "True when identity came out of the learned-ANI regression model", NULL},
    {"_keep", (getter)hit_get_keep, NULL, "the psk_hit records of the query that found the hit (None for a hit built by hand)", NULL},
    {"_idx", (getter)hit_get_idx, NULL, "the hit's record in _keep", NULL}, {NULL, NULL, NULL, NULL, NULL}};
static PyTypeObject HitBase_Type = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "pyskani_amd._hitlist.HitBase", .tp_basicsize = sizeof(HitObject), .tp_dealloc = (destructor)hit_dealloc,
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE, .tp_doc = "storage of pyskani_amd.Hit", .tp_getset = hit_getset, .tp_new = hit_new,
};

/* n psk_hit records -> [cls(...)] * n (cls: a subclass of HitBase without instance storage of its own) */
static PyObject* hit_list(PyTypeObject* tp, const psk_hit* h, Py_ssize_t n, PyObject* qname, PyObject* names, PyObject* keep) {
    const Py_ssize_t n_names = PyList_GET_SIZE(names);
    PyObject* out = PyList_New(n);
    if (!out) return NULL;
    for (Py_ssize_t i = 0; i < n; i++) {
        if ((Py_ssize_t)h[i].ref_index >= n_names) { PyErr_SetString(PyExc_IndexError, "ref_index beyond the database's names"); goto fail; }
        PyObject* ref = PyList_GET_ITEM(names, (Py_ssize_t)h[i].ref_index);
        if (!PyUnicode_Check(ref)) { PyErr_SetString(PyExc_TypeError, "names must be a list of str"); goto fail; }
        HitObject* t = (HitObject*)tp->tp_alloc(tp, 0);
        if (!t) goto fail;
        t->ani = h[i].ani; t->af_query = h[i].af_query; t->af_ref = h[i].af_ref; t->idx = (uint32_t)i; t->learned = h[i].learned ? 1 : 0;
        Py_INCREF(qname); Py_INCREF(ref); Py_INCREF(keep);
        t->qname = qname; t->rname = ref; t->keep = keep;
        PyList_SET_ITEM(out, i, (PyObject*)t);
    }
    return out;
fail:
    Py_DECREF(out);
    return NULL;
}

static int check_cls_names(PyObject* cls, PyObject* names) {
    if (!PyType_Check(cls) || !PyType_IsSubtype((PyTypeObject*)cls, &HitBase_Type) || ((PyTypeObject*)cls)->tp_basicsize != sizeof(HitObject) || ((PyTypeObject*)cls)->tp_itemsize != 0) {
        PyErr_SetString(PyExc_TypeError, "cls must be a subclass of HitBase with empty __slots__"); return -1;
    }
    if (!PyList_Check(names)) { PyErr_SetString(PyExc_TypeError, "names must be a list"); return -1; }
    return 0;
}

/* build(cls, records, n, query_name, names, keep) -> [cls(...)] * n; records: a buffer of n psk_hit */
static PyObject* build(PyObject* self, PyObject* args) {
    PyObject *cls, *recs, *qname, *names, *keep;
    Py_ssize_t n;
    if (!PyArg_ParseTuple(args, "OOnOOO", &cls, &recs, &n, &qname, &names, &keep)) return NULL;
    if (check_cls_names(cls, names)) return NULL;
    Py_buffer view;
    if (PyObject_GetBuffer(recs, &view, PyBUF_SIMPLE) != 0) return NULL;
    if (n < 0 || (size_t)view.len < (size_t)n * sizeof(psk_hit)) { PyBuffer_Release(&view); PyErr_SetString(PyExc_ValueError, "record buffer shorter than n hits"); return NULL; }
    PyObject* out = hit_list((PyTypeObject*)cls, (const psk_hit*)view.buf, n, qname, names, keep);
    PyBuffer_Release(&view);
    return out;
}

/* query_host(fn, free_fn, db, contigs, seed, opts, cls, query_name, names) -> [cls(...)] | int
 * Database.query for host bytes as ONE call from the interpreter (lib.rs:549-660): fn / free_fn are the addresses of the library's psk_query_host / psk_free
 * (the module does not link against the library: the caller resolved them, pyskani_amd/_capi.py), db the psk_db handle, contigs a tuple of bytes objects,
 * opts the address of a psk_query_opts. The interpreter lock is released around the library call (lib.rs:569: py.allow_threads), so queries from several
 * threads overlap; what runs under the lock per call is the argument set-up and the Hit objects, not a ctypes call sequence. A failing call returns its
 * psk_status as an int (the caller raises from the library's thread-local message); the records of a successful one stay reachable as a bytes object behind
 * every Hit (`keep`). */
typedef psk_status (*query_host_fn)(psk_db*, const uint8_t* const*, const uint64_t*, uint32_t, int, const psk_query_opts*, psk_hit**, uint64_t*);
typedef void (*free_fn)(void*);
static PyObject* query_host(PyObject* self, PyObject* args) {
    unsigned long long fn_a, free_a, db_a, opts_a;
    PyObject *contigs, *cls, *qname, *names;
    int seed;
    if (!PyArg_ParseTuple(args, "KKKOiKOOO", &fn_a, &free_a, &db_a, &contigs, &seed, &opts_a, &cls, &qname, &names)) return NULL;
    if (check_cls_names(cls, names)) return NULL;
    if (!PyTuple_Check(contigs)) { PyErr_SetString(PyExc_TypeError, "contigs must be a tuple of bytes"); return NULL; }
    if (!fn_a || !free_a || !db_a || !opts_a) { PyErr_SetString(PyExc_ValueError, "null address"); return NULL; }
    const Py_ssize_t nc = PyTuple_GET_SIZE(contigs);
    const uint8_t* ptr_s[16]; uint64_t len_s[16];
    const uint8_t** ptr = ptr_s; uint64_t* len = len_s;
    if (nc > 16) {
        ptr = (const uint8_t**)PyMem_Malloc(sizeof(*ptr) * (size_t)nc); len = (uint64_t*)PyMem_Malloc(sizeof(*len) * (size_t)nc);
        if (!ptr || !len) { PyMem_Free((void*)ptr); PyMem_Free(len); return PyErr_NoMemory(); }
    }
    PyObject* out = NULL;
    for (Py_ssize_t i = 0; i < nc; i++) {
        PyObject* c = PyTuple_GET_ITEM(contigs, i);      /* (the tuple keeps every bytes object alive for the call) */
        if (!PyBytes_Check(c)) { PyErr_SetString(PyExc_TypeError, "contigs must be a tuple of bytes"); goto done; }
        ptr[i] = (const uint8_t*)PyBytes_AS_STRING(c); len[i] = (uint64_t)PyBytes_GET_SIZE(c);
    }
    {
        psk_hit* hits = NULL; uint64_t n = 0; psk_status rc;
        Py_BEGIN_ALLOW_THREADS
        rc = ((query_host_fn)(uintptr_t)fn_a)((psk_db*)(uintptr_t)db_a, ptr, len, (uint32_t)nc, seed, (const psk_query_opts*)(uintptr_t)opts_a, &hits, &n);
        Py_END_ALLOW_THREADS
        if (rc != PSK_OK) { out = PyLong_FromLong((long)rc); goto done; }
        if (n == 0) out = PyList_New(0);
        else {
            PyObject* keep = PyBytes_FromStringAndSize((const char*)hits, (Py_ssize_t)(n * sizeof(psk_hit)));
            if (keep) { out = hit_list((PyTypeObject*)cls, hits, (Py_ssize_t)n, qname, names, keep); Py_DECREF(keep); }
        }
        if (hits) ((free_fn)(uintptr_t)free_a)(hits);
    }
done:
    if (ptr != ptr_s) { PyMem_Free((void*)ptr); PyMem_Free(len); }
    return out;
}

static PyMethodDef methods[] = {{"build", build, METH_VARARGS, "Hit objects of one query's psk_hit records"},
                                {"query_host", query_host, METH_VARARGS, "psk_query_host on a tuple of bytes -> Hit objects (or the failing psk_status)"}, {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_hitlist", "C-level construction of pyskani_amd.Hit lists", -1, methods};
PyMODINIT_FUNC PyInit__hitlist(void) {
    if (PyType_Ready(&HitBase_Type) < 0) return NULL;
    PyObject* m = PyModule_Create(&moddef);
    if (!m) return NULL;
    Py_INCREF(&HitBase_Type);
    if (PyModule_AddObject(m, "HitBase", (PyObject*)&HitBase_Type) < 0) { Py_DECREF(&HitBase_Type); Py_DECREF(m); return NULL; }
    return m;
}
