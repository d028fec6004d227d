/* fastcall.c -- CPython entry point for the one per-record call of the object API, MinHash.add_sequence.
 *
 * The reference's sketching loop calls add_sequence once per read (src/sourmash/command_sketch.py:746-768 ->
 * src/sourmash/minhash.py:363-371 -> ffi kmerminhash_add_sequence).  Behind it this library only validates the record
 * and queues it (csrc/capi.cpp: add_sequence_dna, smgpu_minhash_add_sequence_rc: 0.1-0.2 us), so for 150-bp reads the
 * binding is what a call costs: through ctypes 0.63-0.70 us (argument conversion 0.36 us of it), i.e. 0.22 Gbase/s from
 * a Python loop.  This module is the same call through the C API: a method descriptor (no bound-method object, no
 * argument tuple), the record's bytes taken in place.
 *
 * It binds nothing at link time: minhash.py hands over the address of smgpu_minhash_add_sequence_rc (taken from the
 * ctypes handle of libsourmash_amd.so it has already loaded) and the Python callable that turns the library's error
 * code into the exception the reference raises.  No hashing happens here and there is no other path to the sketch:
 * without libsourmash_amd.so nothing is bound and add_sequence raises.
 *
 * Build: make -C sourmash_amd/csrc (gcc, Python.h) -> sourmash_amd/_fastcall.<abi>.so, in-tree. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

typedef uint32_t (*add_rc_fn)(void* mh, const char* sequence, uintptr_t len, _Bool force);

static add_rc_fn g_add = NULL;
static PyObject* g_raise = NULL;  /* callable(code): raises the library's last error */
static PyObject* s_objptr = NULL; /* interned "_objptr": the attribute RustObject keeps the native handle in */

static PyObject* type_error(const char* msg) {
    PyErr_SetString(PyExc_TypeError, msg);
    return NULL;
}

/* add_sequence(self, sequence, force=False) -- argument handling of sourmash_amd/minhash.py: to_bytes */
static PyObject* add_sequence(PyObject* self, PyObject* const* args, Py_ssize_t nargs, PyObject* kwnames) {
    PyObject* seq = NULL;
    PyObject* force_obj = NULL;
    if (nargs > 2) return type_error("add_sequence() takes at most 2 positional arguments (sequence, force)");
    if (nargs >= 1) seq = args[0];
    if (nargs == 2) force_obj = args[1];
    if (kwnames) {
        const Py_ssize_t nkw = PyTuple_GET_SIZE(kwnames);
        for (Py_ssize_t i = 0; i < nkw; ++i) {
            PyObject* name = PyTuple_GET_ITEM(kwnames, i);
            PyObject* value = args[nargs + i];
            if (PyUnicode_CompareWithASCIIString(name, "sequence") == 0) {
                if (seq) return type_error("add_sequence() got multiple values for argument 'sequence'");
                seq = value;
            } else if (PyUnicode_CompareWithASCIIString(name, "force") == 0) {
                if (force_obj) return type_error("add_sequence() got multiple values for argument 'force'");
                force_obj = value;
            } else {
                PyErr_Format(PyExc_TypeError, "add_sequence() got an unexpected keyword argument '%U'", name);
                return NULL;
            }
        }
    }
    if (!seq) return type_error("add_sequence() missing 1 required positional argument: 'sequence'");
    int force = 0;
    if (force_obj) {
        force = PyObject_IsTrue(force_obj);
        if (force < 0) return NULL;
    }
    if (!g_add) {
        PyErr_SetString(PyExc_RuntimeError, "sourmash_amd._fastcall is not bound to libsourmash_amd.so");
        return NULL;
    }
    /* the native handle (utils.py: RustObject._get_objptr) */
    PyObject* po = PyObject_GetAttr(self, s_objptr);
    if (!po) return NULL;
    void* handle = NULL;
    if (po != Py_None) {
        handle = PyLong_AsVoidPtr(po);
        if (!handle && PyErr_Occurred()) {
            Py_DECREF(po);
            return NULL;
        }
    }
    Py_DECREF(po);
    if (!handle) {
        PyErr_SetString(PyExc_RuntimeError, "Object is closed");
        return NULL;
    }
    /* the record's bytes, in place */
    const char* buf = NULL;
    Py_ssize_t len = 0;
    Py_buffer view;
    int have_view = 0;
    char one;
    if (PyBytes_Check(seq)) {
        buf = PyBytes_AS_STRING(seq);
        len = PyBytes_GET_SIZE(seq);
    } else if (PyUnicode_Check(seq)) {
        buf = PyUnicode_AsUTF8AndSize(seq, &len); /* the bytes of str.encode("utf-8"), cached in the object */
        if (!buf) return NULL;
    } else if (PyLong_Check(seq)) {
        const long v = PyLong_AsLong(seq);
        if (v == -1 && PyErr_Occurred()) return NULL;
        if (v < 0 || v > 255) {
            PyErr_SetString(PyExc_ValueError, "bytes must be in range(0, 256)");
            return NULL;
        }
        one = (char)v;
        buf = &one;
        len = 1;
    } else if (PyByteArray_Check(seq) || PyMemoryView_Check(seq)) {
        if (PyObject_GetBuffer(seq, &view, PyBUF_SIMPLE) != 0) return NULL;
        have_view = 1;
        buf = (const char*)view.buf;
        len = view.len;
    } else {
        return type_error("Requires a string-like sequence");
    }
    const uint32_t code = g_add(handle, buf, (uintptr_t)len, force ? 1 : 0);
    if (have_view) PyBuffer_Release(&view);
    if (code) {
        PyObject* r = PyObject_CallFunction(g_raise, "I", (unsigned int)code);
        if (r) { /* the hook must raise */
            Py_DECREF(r);
            PyErr_Format(PyExc_RuntimeError, "libsourmash_amd error code %u", (unsigned int)code);
        }
        return NULL;
    }
    Py_RETURN_NONE;
}

PyDoc_STRVAR(add_sequence_doc,
             "add_sequence(sequence, force=False)\n--\n\n"
             "Add every k-mer of a DNA sequence (GPU).  The record is validated and queued; the library hashes the queue in\n"
             "one kernel launch when it is large or when the sketch is next looked at, so a loop over reads costs one C call\n"
             "per read and no launch.  Invalid DNA with force=False raises here, after the k-mers in front of the bad one\n"
             "were queued -- the reference's streaming order (signature.rs:48-54).");

static PyMethodDef base_methods[] = {
    {"add_sequence", (PyCFunction)(void (*)(void))add_sequence, METH_FASTCALL | METH_KEYWORDS, add_sequence_doc},
    {NULL, NULL, 0, NULL},
};

/* no fields, no tp_new of its own: layout and construction are object's, so that it can sit next to RustObject in the
 * bases of MinHash (object.__new__(cls) in RustObject._from_objptr stays legal) */
static PyTypeObject AddSequenceBase = {
    PyVarObject_HEAD_INIT(NULL, 0)
    .tp_name = "sourmash_amd._fastcall.AddSequenceBase",
    .tp_basicsize = sizeof(PyObject),
    .tp_flags = Py_TPFLAGS_DEFAULT | Py_TPFLAGS_BASETYPE,
    .tp_doc = "Mix-in that gives MinHash its add_sequence as a C method (see csrc/fastcall.c).",
    .tp_methods = base_methods,
};

/* bind(address of smgpu_minhash_add_sequence_rc, callable(code) that raises) */
static PyObject* bind(PyObject* module, PyObject* args) {
    (void)module;
    PyObject* addr = NULL;
    PyObject* hook = NULL;
    if (!PyArg_ParseTuple(args, "OO", &addr, &hook)) return NULL;
    void* p = PyLong_AsVoidPtr(addr);
    if (!p) {
        if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "null entry point");
        return NULL;
    }
    if (!PyCallable_Check(hook)) return type_error("the error hook must be callable");
    Py_INCREF(hook);
    Py_XSETREF(g_raise, hook);
    g_add = (add_rc_fn)p;
    Py_RETURN_NONE;
}

static PyMethodDef module_methods[] = {
    {"bind", bind, METH_VARARGS, "bind(entry_address, raise_hook): connect add_sequence to the loaded libsourmash_amd.so"},
    {NULL, NULL, 0, NULL},
};

static struct PyModuleDef moduledef = {
    PyModuleDef_HEAD_INIT, "sourmash_amd._fastcall", "C-API entry of MinHash.add_sequence (csrc/fastcall.c)", -1, module_methods,
    NULL, NULL, NULL, NULL,
};

PyMODINIT_FUNC PyInit__fastcall(void) {
    s_objptr = PyUnicode_InternFromString("_objptr");
    if (!s_objptr) return NULL;
    if (PyType_Ready(&AddSequenceBase) < 0) return NULL;
    PyObject* m = PyModule_Create(&moduledef);
    if (!m) return NULL;
    Py_INCREF(&AddSequenceBase);
    if (PyModule_AddObject(m, "AddSequenceBase", (PyObject*)&AddSequenceBase) < 0) {
        Py_DECREF(&AddSequenceBase);
        Py_DECREF(m);
        return NULL;
    }
    return m;
}
