"""
ref_loader.py -- import the *real* reference (zharmad/SpinRelax at /root/reference) in the build
container.  TEST INFRASTRUCTURE ONLY; used by oracle/gen_golden.py and by tests that are skipped
when /root/reference is absent (it never exists on the GPU box).

Recipe:
  * oracle/_ref/npufunc.so is Jomega/Jomega.c compiled by oracle/Makefile (gcc, no numpy.distutils);
  * mdtraj / transforms3d are not installed: empty stub modules are registered (no hot-path function
    touches them);
  * the hyphenated scripts are executed with importlib and their trailing module-level sys.exit()
    is caught.
Nothing is copied from the reference; it is imported from where it lies.
"""
import importlib.util
import os
import sys
import types

REF = os.environ.get('SPINRELAX_REFERENCE', '/root/reference')
_HERE = os.path.dirname(os.path.abspath(__file__))
_cache = {}


def available():
    return os.path.isfile(os.path.join(REF, 'spectral_densities.py')) and \
        os.path.isfile(os.path.join(_HERE, '_ref', 'npufunc.so'))


def _prepare():
    for p in (os.path.join(_HERE, '_ref'), REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    for m in ('mdtraj', 'transforms3d', 'transforms3d.quaternions'):
        sys.modules.setdefault(m, types.ModuleType(m))


def _load_script(name, fn):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, fn))
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except SystemExit:
        pass
    return mod


def load():
    """Returns a namespace with the reference modules: sd, fitCt, gm, gs, qs, npufunc, calcCt, calcRelax."""
    if 'ns' in _cache:
        return _cache['ns']
    if not available():
        raise RuntimeError('reference not available (needs %s and oracle/_ref/npufunc.so; run `make -C oracle ref`)' % REF)
    _prepare()
    import npufunc
    import spectral_densities as sd
    import fitting_Ct_functions as fitCt
    import general_maths as gm
    import general_scripts as gs
    import transforms3d_supplement as qs
    ns = types.SimpleNamespace(npufunc=npufunc, sd=sd, fitCt=fitCt, gm=gm, gs=gs, qs=qs)
    ns.calcCt = _load_script('ref_calcCt', 'calculate-Ct-from-traj.py')
    ns.calcRelax = _load_script('ref_calcRelax', 'calculate-relaxations-from-Ct.py')
    _cache['ns'] = ns
    return ns
