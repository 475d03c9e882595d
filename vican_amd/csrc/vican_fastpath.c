/* vican_fastpath.c - the drop-in front-end's passes over the caller's edge dict, in C (CPython + NumPy C API).
 *
 * The reference hands the solver a dict of Python objects ({(cam, "<t>_<marker>"): {"pose": SE3, "corners": ..., ...}},
 * bipgo.py:203-221); once the solve takes 2 ms the drop-in call is bound by the interpreter touching 80 000 pose objects a
 * handful of times (profiles/r05_frontend_phases.txt: pose objects 2.4 ms, R() 3.1, stacking 4.8, t() 3.4, stacking 3.2 - per
 * column one list comprehension and one join).  These functions make ONE pass per column and copy straight into the output
 * array.  Host-side plumbing only: vican_amd.frontend falls back to its NumPy / list-comprehension path when this module is
 * missing or returns None (anything it does not recognise: non-dict values, other pose classes' oddities, other dtypes).
 *
 * Built by vican_amd._lib.build_fastpath() (gcc, in-tree: vican_amd/csrc/_vican_fastpath.so).                                */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#define NPY_NO_DEPRECATED_API NPY_1_7_API_VERSION
#include <numpy/arrayobject.h>
#include <string.h>

/* copy the n_elem values of array `a` (float64 or float32, any strides, exactly n_elem elements) to dst as float64, in C order;
 * returns 0 = float64 source, 1 = float32 source, -1 = not recognised */
static int copy_small(PyObject* obj, double* dst, npy_intp n_elem) {
    if (!PyArray_Check(obj)) return -1;
    PyArrayObject* a = (PyArrayObject*)obj;
    if (PyArray_SIZE(a) != n_elem) return -1;
    const int tp = PyArray_TYPE(a);
    if (tp != NPY_FLOAT64 && tp != NPY_FLOAT32) return -1;
    if (!PyArray_ISALIGNED(a) || !PyArray_ISNOTSWAPPED(a)) return -1;
    const char* base = (const char*)PyArray_DATA(a);
    if (tp == NPY_FLOAT64 && PyArray_IS_C_CONTIGUOUS(a)) {
        memcpy(dst, base, (size_t)n_elem * sizeof(double));
        return 0;
    }
    const int nd = PyArray_NDIM(a);
    if (nd > 2) return -1;
    const npy_intp* sh = PyArray_SHAPE(a);
    const npy_intp* st = PyArray_STRIDES(a);
    const npy_intp n0 = nd == 2 ? sh[0] : 1, n1 = nd >= 1 ? sh[nd - 1] : 1;
    const npy_intp s0 = nd == 2 ? st[0] : 0, s1 = nd >= 1 ? st[nd - 1] : 0;
    npy_intp k = 0;
    for (npy_intp i = 0; i < n0; ++i)
        for (npy_intp j = 0; j < n1; ++j) {
            const char* p = base + i * s0 + j * s1;
            dst[k++] = tp == NPY_FLOAT64 ? *(const double*)p : (double)*(const float*)p;
        }
    return tp == NPY_FLOAT32 ? 1 : 0;
}

/* gather_poses(vals, se3_type) -> (R [n,3,3] float64, t [n,3] float64, r_is_f32 [n] uint8) or None
 * vals: list of dicts with a "pose" entry; poses of exactly `se3_type` (vican_amd.geometry.SE3, whose R() / t() return the
 * attributes _R / _t) are read as attributes, any other pose through its R() and t() methods. */
static PyObject* gather_poses(PyObject* self, PyObject* args) {
    PyObject *vals, *se3_type;
    if (!PyArg_ParseTuple(args, "OO", &vals, &se3_type)) return NULL;
    if (!PyList_CheckExact(vals)) Py_RETURN_NONE;
    const Py_ssize_t n = PyList_GET_SIZE(vals);
    npy_intp dR[3] = {n, 3, 3}, dt[2] = {n, 3}, df[1] = {n};
    PyObject* R = PyArray_SimpleNew(3, dR, NPY_FLOAT64);
    PyObject* t = PyArray_SimpleNew(2, dt, NPY_FLOAT64);
    PyObject* f = PyArray_SimpleNew(1, df, NPY_UINT8);
    PyObject *k_pose = PyUnicode_InternFromString("pose"), *k_R = PyUnicode_InternFromString("_R"), *k_t = PyUnicode_InternFromString("_t"),
             *m_R = PyUnicode_InternFromString("R"), *m_t = PyUnicode_InternFromString("t");
    int ok = R && t && f && k_pose && k_R && k_t && m_R && m_t;
    double* pR = ok ? (double*)PyArray_DATA((PyArrayObject*)R) : NULL;
    double* pt = ok ? (double*)PyArray_DATA((PyArrayObject*)t) : NULL;
    unsigned char* pf = ok ? (unsigned char*)PyArray_DATA((PyArrayObject*)f) : NULL;
    int unknown = 0;
    for (Py_ssize_t i = 0; ok && !unknown && i < n; ++i) {
        PyObject* v = PyList_GET_ITEM(vals, i);
        if (!PyDict_CheckExact(v)) { unknown = 1; break; }
        PyObject* pose = PyDict_GetItemWithError(v, k_pose);            /* borrowed */
        if (!pose) { if (PyErr_Occurred()) ok = 0; else unknown = 1; break; }
        PyObject *ro, *to;
        if ((PyObject*)Py_TYPE(pose) == se3_type) {
            ro = PyObject_GetAttr(pose, k_R);
            to = ro ? PyObject_GetAttr(pose, k_t) : NULL;
        } else {
            ro = PyObject_CallMethodNoArgs(pose, m_R);
            to = ro ? PyObject_CallMethodNoArgs(pose, m_t) : NULL;
        }
        if (!ro || !to) { Py_XDECREF(ro); Py_XDECREF(to); ok = 0; break; }
        const int kr = copy_small(ro, pR + 9 * i, 9), kt = copy_small(to, pt + 3 * i, 3);
        Py_DECREF(ro); Py_DECREF(to);
        if (kr < 0 || kt < 0) { unknown = 1; break; }
        pf[i] = (unsigned char)kr;
    }
    Py_XDECREF(k_pose); Py_XDECREF(k_R); Py_XDECREF(k_t); Py_XDECREF(m_R); Py_XDECREF(m_t);
    if (!ok || unknown) {
        Py_XDECREF(R); Py_XDECREF(t); Py_XDECREF(f);
        if (!ok) return NULL;                                            /* a Python error is set: the caller sees it */
        Py_RETURN_NONE;
    }
    PyObject* out = PyTuple_Pack(3, R, t, f);
    Py_DECREF(R); Py_DECREF(t); Py_DECREF(f);
    return out;
}

/* gather_item(vals, key, n_elem) -> float64 array [n, n_elem] of v[key] (arrays of n_elem float64 / float32 values, or - for
 * n_elem == 1 - Python / NumPy scalars), or None */
static PyObject* gather_item(PyObject* self, PyObject* args) {
    PyObject *vals, *key;
    Py_ssize_t n_elem;
    if (!PyArg_ParseTuple(args, "OOn", &vals, &key, &n_elem)) return NULL;
    if (!PyList_CheckExact(vals) || n_elem < 1) Py_RETURN_NONE;
    const Py_ssize_t n = PyList_GET_SIZE(vals);
    npy_intp d[2] = {n, n_elem};
    PyObject* out = PyArray_SimpleNew(2, d, NPY_FLOAT64);
    if (!out) return NULL;
    double* p = (double*)PyArray_DATA((PyArrayObject*)out);
    for (Py_ssize_t i = 0; i < n; ++i) {
        PyObject* v = PyList_GET_ITEM(vals, i);
        if (!PyDict_CheckExact(v)) { Py_DECREF(out); Py_RETURN_NONE; }
        PyObject* x = PyDict_GetItemWithError(v, key);
        if (!x) {
            Py_DECREF(out);
            if (PyErr_Occurred()) return NULL;
            Py_RETURN_NONE;                                              /* (the Python path raises the KeyError) */
        }
        if (copy_small(x, p + (size_t)i * n_elem, n_elem) >= 0) continue;
        if (n_elem == 1 && (PyFloat_Check(x) || PyLong_Check(x))) {
            const double s = PyFloat_AsDouble(x);
            if (s == -1.0 && PyErr_Occurred()) { Py_DECREF(out); return NULL; }
            p[i] = s;
            continue;
        }
        Py_DECREF(out);
        Py_RETURN_NONE;
    }
    return out;
}

static PyMethodDef methods[] = {
    {"gather_poses", gather_poses, METH_VARARGS, "R [n,3,3], t [n,3], float32 flags [n] of the poses of a list of edge value dicts"},
    {"gather_item", gather_item, METH_VARARGS, "float64 [n, n_elem] of v[key] over a list of edge value dicts"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_vican_fastpath", "C passes of the vican_amd front-end", -1, methods};
PyMODINIT_FUNC PyInit__vican_fastpath(void) {
    import_array();
    return PyModule_Create(&moddef);
}
