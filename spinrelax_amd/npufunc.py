"""
Stand-in for the reference's only native symbol, the numpy ufunc `npufunc.Jomega` (Jomega/Jomega.c:30-156), evaluated on
the GPU through sr_jomega_f64:  Jomega(x, y) = x / (x*x + y*y)  elementwise with numpy broadcasting, plus the `.outer`
form the reference actually calls (`npufunc.Jomega.outer(D_J, omega)`, spectral_densities.py:1971).

    import spinrelax_amd.npufunc as npufunc         # instead of the compiled module
    npufunc.Jomega.outer(D_J, om)                    # (3,) x (5,) -> (3, 5) float64

The reference registers four type loops, `ee->e`, `ff->f`, `dd->d`, `gg->g` (Jomega.c:107-156).  The result dtype here
follows the same rule (the common dtype of the inputs among half / single / double / long double, anything else through
double); the arithmetic is done in float64 on the device and rounded once to that dtype.  Two deliberate differences:
the reference's half loop reads the raw float16 bits as if they were floats (Jomega.c:98-100, never reached by a caller)
-- here half inputs are converted properly; and long double is evaluated in float64.
"""
import numpy as np

from . import hip


class _JomegaUfunc:
    nin = 2
    nout = 1
    nargs = 3
    types = ['ee->e', 'ff->f', 'dd->d', 'gg->g']
    __name__ = 'Jomega'

    def __init__(self):
        self._ctx = None

    def _context(self):
        return self._ctx if self._ctx is not None else hip.default_context()

    def bind(self, ctx):
        """Use a specific spinrelax_amd.hip.Context instead of the process-wide default."""
        self._ctx = ctx
        return self

    @staticmethod
    def _out_dtype(x, y):
        dt = np.result_type(x, y)
        for cand in (np.float16, np.float32, np.float64, np.longdouble):
            if dt == np.dtype(cand):
                return dt
        if np.can_cast(dt, np.float64):
            return np.dtype(np.float64)
        raise TypeError("ufunc 'Jomega' not supported for the input types %s" % dt)

    def __call__(self, x, y):
        x = np.asarray(x)
        y = np.asarray(y)
        dt = self._out_dtype(x, y)
        xb, yb = np.broadcast_arrays(x.astype(np.float64), y.astype(np.float64))
        if xb.size == 0:
            return np.empty(xb.shape, dtype=dt)
        out = self._context().jomega(np.ascontiguousarray(xb), np.ascontiguousarray(yb)).astype(dt, copy=False)
        return out[()] if out.ndim == 0 else out

    def outer(self, a, b):
        a = np.asarray(a)
        b = np.asarray(b)
        return self(a.reshape(a.shape + (1,) * b.ndim), b)


Jomega = _JomegaUfunc()
