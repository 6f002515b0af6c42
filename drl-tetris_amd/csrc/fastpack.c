/* _fastpack — host-side glue of the drop-in Python API (drl-tetris_amd/environment.py), NOT part of the hot path's compute:
 * turns the reference's calling convention for perform_action — a Python list of `action` objects, each a list of key ints
 * (environment/data_types/action.py:2-9, tetris_environment_vector.py:90-97) — into the uint8 arrays tetris_step_keys takes,
 * in one pass in C.  In pure Python the same packing is ~0.5 ms per 4 096 actions (bytes() of every list + a join + a scatter),
 * the largest single item of a worker-loop iteration; here it is ~0.1 ms.  Built by __graft_entry__.build() with gcc; where the
 * module is missing environment.py uses its Python packer (same result).
 *
 *   max_len(actions, action_type) -> int          longest action; TypeError if an element is not exactly `action_type`
 *   fill(actions, who_addr, keys_addr, lens_addr, n, P, K) -> None
 *        who  int64 [n]: the acting player of each env;  keys uint8 [n][P][K] (zeroed by the caller);  lens uint8 [n][P]
 *        (set to 1 by the caller: the other players get the null action [0]).  ValueError for a key outside 0..255.      */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

static PyObject *fp_max_len(PyObject *self, PyObject *args) {
    PyObject *actions, *type;
    if (!PyArg_ParseTuple(args, "O!O", &PyList_Type, &actions, &type)) return NULL;
    Py_ssize_t n = PyList_GET_SIZE(actions), best = 0;
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject *a = PyList_GET_ITEM(actions, i);
        if ((PyObject *)Py_TYPE(a) != type) {
            PyErr_Format(PyExc_TypeError, "perform_action(action a, int p) was called with type(action)=%s", Py_TYPE(a)->tp_name);
            return NULL;
        }
        Py_ssize_t len = PyList_GET_SIZE(a);
        if (len > best) best = len;
    }
    return PyLong_FromSsize_t(best);
}

static PyObject *fp_fill(PyObject *self, PyObject *args) {
    PyObject *actions;
    unsigned long long who_addr, keys_addr, lens_addr;
    Py_ssize_t n, P, K;
    if (!PyArg_ParseTuple(args, "O!KKKnnn", &PyList_Type, &actions, &who_addr, &keys_addr, &lens_addr, &n, &P, &K)) return NULL;
    if (PyList_GET_SIZE(actions) != n || P < 1 || K < 1) { PyErr_SetString(PyExc_ValueError, "fill: n / P / K"); return NULL; }
    const int64_t *who = (const int64_t *)(uintptr_t)who_addr;
    uint8_t *keys = (uint8_t *)(uintptr_t)keys_addr, *lens = (uint8_t *)(uintptr_t)lens_addr;
    for (Py_ssize_t i = 0; i < n; i++) {
        PyObject *a = PyList_GET_ITEM(actions, i);
        if (!PyList_Check(a)) { PyErr_SetString(PyExc_TypeError, "an action must be a list of key ints"); return NULL; }
        Py_ssize_t len = PyList_GET_SIZE(a);
        int64_t p = who[i];
        if (len > K || len > 255 || p < 0 || p >= P) { PyErr_SetString(PyExc_ValueError, "fill: action longer than K / 255 keys, or player out of range"); return NULL; }
        uint8_t *dst = keys + ((size_t)i * (size_t)P + (size_t)p) * (size_t)K;
        for (Py_ssize_t k = 0; k < len; k++) {
            long v = PyLong_AsLong(PyList_GET_ITEM(a, k));
            if (v < 0 || v > 255) {
                if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "bytes must be in range(0, 256)");
                return NULL;
            }
            dst[k] = (uint8_t)v;
        }
        lens[(size_t)i * (size_t)P + (size_t)p] = (uint8_t)len;
    }
    Py_RETURN_NONE;
}

static PyMethodDef methods[] = {
    {"max_len", fp_max_len, METH_VARARGS, "longest action of a list; checks the element type"},
    {"fill", fp_fill, METH_VARARGS, "write the actions' keys and lengths into the caller's arrays"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_fastpack", "list-of-actions -> uint8 arrays (drop-in API glue)", -1, methods};
PyMODINIT_FUNC PyInit__fastpack(void) { return PyModule_Create(&module); }
