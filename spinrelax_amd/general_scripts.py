"""
Text / numpy file formats at the drop-in boundary: the files run-all.bash passes
between the steps of the workflow.  Writers produce byte-identical text to the reference's
general_scripts.py for the same arrays (tests/test_formats.py compares against files written by the
reference itself); readers accept what the reference's writers produce.

Reference: general_scripts.py:11-16, 47-67, 182-290; calculate-Ct-from-traj.py:604-630.
"""
import sys

import os

import numpy as np


def normalise_vector_array(v):
    """general_scripts.py:11-16."""
    return v / np.sqrt((v ** 2).sum(-1))[..., np.newaxis]


def _is_data(line):
    return not (line == "" or line[0] in "#@&" or line[0] == "\n")


def load_xy(fn):
    """general_scripts.py:47-56: two-column file, comment / xmgrace lines skipped."""
    x, y = [], []
    for l in open(fn):
        if not _is_data(l):
            continue
        w = l.split()
        x.append(float(w[0]))
        y.append(float(w[1]))
    return np.array(x), np.array(y)


def load_xys(fn):
    """general_scripts.py:58-67: first column x, remaining columns y."""
    x, y = [], []
    for l in open(fn):
        if l == "" or l[0] in "#@&" or l.strip() == "":
            continue
        v = [float(i) for i in l.split()]
        x.append(v[0])
        y.append(v[1:])
    return np.array(x), np.array(y)


def _native_lib():
    """the shared library's host-side text routines (csrc/sr_textio.hip), or None: formatting and parsing text is not device
    work, so without the library the Python implementations below do the same job (slower)"""
    try:
        from . import _lib
        return _lib.load()
    except Exception:
        return None


def _load_sxydylist_native(fn, key):
    """load_sxydylist through sr_text_open_sxydy (threaded strtod parse); None when the file is not the regular case"""
    lib = _native_lib()
    if lib is None:
        return None
    import ctypes
    h = lib.sr_text_open_sxydy(os.fsencode(fn), key.encode(), 0)
    if not h:
        return None
    try:
        info = (ctypes.c_int64 * 6)()
        if lib.sr_text_sxydy_info(h, info) != 0 or not info[0] or info[1] == 0:
            return None
        nsets, npts, has_dy, nbytes = info[1], info[2], info[3], info[5]
        x = np.empty((nsets, npts))
        y = np.empty((nsets, npts))
        dy = np.empty((nsets, npts)) if has_dy else None
        lb = ctypes.create_string_buffer(max(1, nbytes))
        if lib.sr_text_sxydy_get(h, x.ctypes.data, y.ctypes.data, dy.ctypes.data if has_dy else None, lb) != 0:
            return None
        legs = [t.decode() for t in lb.raw[:nbytes].split(b'\0')[:info[4]]]
        return legs, x, y, (dy if has_dy else [])
    finally:
        lib.sr_text_close_sxydy(h)


def load_sxydylist(fn, key="legend"):
    """general_scripts.py:182-213: xmgrace multi-set file with `@s<i> legend "<name>"` lines; sets end at '&'.
    Returns (legends, x[nset, npts], y, dy) -- dy is [] when the file has no third column.  Regular files (every set the same
    length, two or three plain numeric columns) are parsed natively; everything else line by line as the reference does."""
    got = _load_sxydylist_native(fn, key)
    if got is not None:
        return got
    legs, xs, ys, dys = [], [], [], []
    x, y, dy = [], [], []
    for l in open(fn):
        if l == "" or l == "\n":
            continue
        w = l.split()
        if l[0] == "#" or l[0] == "@":
            if key in l:
                legs.append(w[-1].strip('"'))
            continue
        if l[0] == "&":
            xs.append(x)
            ys.append(y)
            if len(dy) > 0:
                dys.append(dy)
            x, y, dy = [], [], []
            continue
        x.append(float(w[0]))
        y.append(float(w[1]))
        if len(w) > 2:
            dy.append(float(w[2]))
    if x != []:
        xs.append(x)
        ys.append(y)
        dys.append(dy)
    if dys != []:
        return legs, np.array(xs), np.array(ys), np.array(dys)
    return legs, np.array(xs), np.array(ys), []


def print_xy(fn, x, y, dy=None, header=""):
    """general_scripts.py:231-244: `print(x[i], y[i][, dy[i]])` per line, i.e. the str() of the numpy
    scalars (float32 for the relaxation datablock).  dy=None/empty -> two columns."""
    with open(fn, 'w') as fp:
        if header != "":
            print(header, file=fp)
        if dy is None or len(dy) == 0:
            for i in range(len(x)):
                print(x[i], y[i], file=fp)
        else:
            for i in range(len(x)):
                print(x[i], y[i], dy[i], file=fp)


def print_xydy(fn, x, y, dy, header=""):
    print_xy(fn, x, y, dy, header)


def print_xylist(fn, x, ylist, bCols=False, header=""):
    """general_scripts.py:246-273 (x(nvals), y(nplots, nvals); bCols stacks the plots as columns, %g)."""
    ylist = np.array(ylist)
    with open(fn, 'w') as fp:
        if header != "":
            print(header, file=fp)
        if ylist.ndim == 1:
            for j in range(len(x)):
                print(x[j], ylist[j], file=fp)
            print("&", file=fp)
        elif ylist.ndim == 2:
            nplot, nvals = ylist.shape
            if bCols:
                for j in range(nvals):
                    print("%g " % x[j] + " ".join("%g" % ylist[i][j] for i in range(nplot)), file=fp)
                print("&", file=fp)
            else:
                for i in range(nplot):
                    for j in range(len(x)):
                        print(x[j], ylist[i][j], file=fp)
                    print("&", file=fp)


def _numpy_str_pairs(y):
    """str(row).strip('[]') for every row of a float64 array of shape (n, 2), as numpy's array printer formats a
    two-element float64 vector (numpy/_core/arrayprint.py FloatingFormat, default print options: precision 8, floatmode
    'maxprec') -- but for all rows at once instead of one array2string call per line (which is what made writing the two
    C(t) files of a 512-residue run take 30 s).  Per row: scientific notation when the largest magnitude is >= 1e8, the
    smallest non-zero one < 1e-4 or their ratio > 1000, positional otherwise; digits = shortest round-trip digits, at
    most 8 after the point, trailing zeros dropped; both elements padded to common widths on either side of the point.
    Rows this function is not sure about (non-finite values, three-digit exponents) are formatted by numpy itself.
    tests/test_formats_and_hostlogic.py compares it with numpy's own output on a few hundred thousand pairs."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    n = y.shape[0]
    out = np.empty(n, dtype=object)
    a = np.abs(y)
    finite = np.isfinite(y).all(axis=1)
    nz = a > 0
    big = np.where(nz, a, 0.0).max(axis=1)
    small = np.where(nz, a, np.inf).min(axis=1)
    with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
        expf = nz.any(axis=1) & ((big >= 1e8) | (small < 1e-4) | (big / small > 1000.0))
    slow = ~finite | (expf & ((big >= 1e100) | (small < 1e-99)))
    pos = finite & ~expf & ~slow
    if pos.any():
        idx = np.nonzero(pos)[0]
        s8 = np.char.rstrip(np.char.mod('%.8f', y[idx]), '0')                 # '0.12345678', '12.', '-0.5'
        dot = np.char.find(s8, '.')
        ln = np.char.str_len(s8)
        frac = ln - dot - 1
        padl = dot.max(axis=1, keepdims=True) - dot
        padr = frac.max(axis=1, keepdims=True) - frac
        e = np.char.ljust(np.char.rjust(s8, ln + padl), ln + padl + padr)
        out[idx] = np.char.add(np.char.add(e[:, 0], ' '), e[:, 1])
    ex = finite & expf & ~slow
    if ex.any():
        idx = np.nonzero(ex)[0]
        s8 = np.char.mod('%.8e', y[idx])                                       # '-1.23400000e-05'
        epos = np.char.find(s8, 'e')
        mant = np.char.rstrip(np.array([[u[:k] for u, k in zip(r, kk)] for r, kk in zip(s8, epos)]), '0')
        dot = np.char.find(mant, '.')
        prec = (np.char.str_len(mant) - dot - 1).max(axis=1)                   # digits after the point, common to the row
        rows = np.empty(idx.size, dtype=object)
        for p in np.unique(prec):
            sel = np.nonzero(prec == p)[0]
            sp = np.char.mod('%%.%de' % p, y[idx[sel]])
            if p == 0:                                                         # C prints '1e-05', numpy keeps the point: '1.e-05'
                sp = np.char.replace(sp, 'e', '.e')
            d2 = np.char.find(sp, '.')
            padl = d2.max(axis=1, keepdims=True) - d2
            e = np.char.rjust(sp, np.char.str_len(sp) + padl)
            rows[sel] = np.char.add(np.char.add(e[:, 0], ' '), e[:, 1])
        out[idx] = rows
    for j in np.nonzero(slow)[0]:
        out[j] = str(y[j]).strip('[]')
    return out


def print_sxylist(fn, legend, x, ylist, header=[]):
    """general_scripts.py:275-290 -- the `_Ctint.dat` / `_Ctext.dat` writer: per set `@s<i> legend "<name>"`,
    then `x[j] <numpy str of ylist[i][j] without brackets>`, then `&`.  The numeric text is whatever numpy's
    str() gives for the array dtype (8 significant digits), exactly like the reference; (n, 2) float64 sets (C(t) with its
    error) are formatted a whole set at a time (_numpy_str_pairs), anything else line by line through numpy."""
    ylist = np.array(ylist)
    xs = [str(v) for v in x]
    fast = ylist.dtype == np.float64 and ylist.ndim == 3 and ylist.shape[2] == 2
    if fast and len(xs) == ylist.shape[1]:
        # the same bytes from the library's threaded formatter (csrc/sr_textio.hip); rows outside the regular formats make it
        # decline (return code 1) and the code below writes the file
        lib = _native_lib()
        if lib is not None:
            y = np.ascontiguousarray(ylist, dtype=np.float64)
            legs = b''.join(('@s%d legend "%s"' % (i, legend[i])).encode() + b'\0' for i in range(y.shape[0]))
            xsb = b''.join(v.encode() + b'\0' for v in xs)
            hdr = ''.join('%s\n' % line for line in header).encode()
            rc = lib.sr_text_write_sxydy_f64(os.fsencode(fn), hdr, y.shape[0], y.shape[1], legs, xsb, y.ctypes.data, 0)
            if rc == 0:
                return
            if rc < 0:
                raise IOError('cannot write %s' % fn)
    with open(fn, 'w') as fp:
        for line in header:
            print("%s" % line, file=fp)
        for i in range(len(ylist)):
            print("@s%d legend \"%s\"" % (i, legend[i]), file=fp)
            if fast:
                ys = _numpy_str_pairs(ylist[i])
                fp.write(''.join('%s %s\n' % (a, b) for a, b in zip(xs, ys)))
            else:
                for j in range(len(x)):
                    print(x[j], str(ylist[i][j]).strip('[]'), file=fp)
            print("&", file=fp)


def print_s3d(fn, legend, arr, cols, header=[]):
    """general_scripts.py:292-307 (text form of --vecDist)."""
    with open(fn, 'w') as fp:
        for line in header:
            print("%s" % line, file=fp)
        for i in range(arr.shape[0]):
            print("@s%d legend \"%s\"" % (i, legend[i]), file=fp)
            for j in range(arr.shape[1]):
                print(" ".join("%g" % arr[i, j, c] for c in cols), file=fp)
            print("&", file=fp)


def print_gplot_hist(fn, hist, edges, header='', bSphere=False):
    """general_scripts.py:327-381: gnuplot-style bin-centre listing (sphere completion rows when bSphere)."""
    nbins = hist.shape
    dim = len(nbins)
    with open(fn, 'w') as fp:
        if header != '':
            print('%s' % header, file=fp)
        print('# DIMENSIONS: %i' % dim, file=fp)
        print("# BINWIDTH: " + " ".join("%g" % ((edges[i][-1] - edges[i][0]) / nbins[i]) for i in range(dim)), file=fp)
        print("# NBINS: " + " ".join("%g" % (nbins[i]) for i in range(dim)), file=fp)
        if bSphere:
            if dim != 2:
                print("= = = ERROR: histogram data is not in 2D, but spherical histogram plotting is requested!", file=sys.stderr)
                sys.exit(1)
            xmin = 0.5 * (edges[0][0] + edges[0][1])
            ymin, ymax = edges[1][0], edges[1][-1]
            rows = [(0.5 * (edges[0][eX] + edges[0][eX + 1]), hist[eX]) for eX in range(nbins[0])]
            rows.append((xmin + 2 * np.pi, hist[0]))
            for xavg, col in rows:
                print('%g %g %g' % (xavg, ymin, col[0]), file=fp)
                for eY in range(nbins[1]):
                    print('%g %g %g' % (xavg, 0.5 * (edges[1][eY] + edges[1][eY + 1]), col[eY]), file=fp)
                print('%g %g %g' % (xavg, ymax, col[-1]), file=fp)
                print('', file=fp)
        else:
            for index, val in np.ndenumerate(hist):
                s = " ".join("%g" % (0.5 * (edges[i][index[i]] + edges[i][index[i] + 1])) for i in range(dim))
                print(s + " %g" % val, file=fp)
                if index[-1] == nbins[-1] - 1:
                    print('', file=fp)


def save_vecHistogram_npz(fn, names, hist, edges):
    """calculate-Ct-from-traj.py:629-630.  numpy >= 1.24 refuses the ragged `edges` list the reference
    passes; it is stored as the object array older numpy built implicitly (readers use allow_pickle and
    index edges[0], edges[1]: spectral_densities.py:285-292, 2338-2339)."""
    e = np.empty(2, dtype=object)
    e[0], e[1] = edges[0], edges[1]
    np.savez_compressed(fn, names=names, dataType='LambertCylindrical', bHistogram=True, edges=e,
                        axisLabels=['phi', 'cos(theta)'], data=hist)


def save_vecPhiTheta_npz(fn, names, phitheta):
    """calculate-Ct-from-traj.py:604-605."""
    np.savez_compressed(fn, names=names, dataType='PhiTheta', axisLabels=['phi', 'theta'], bHistogram=False, data=phitheta)
