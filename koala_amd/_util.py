"""Locations of the native library and the default KNS1 model (mirrors reference binding/python/_util.py:59-84)."""

import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))


def default_library_path(relative: str = '') -> str:
    """The in-tree HIP build of libpv_koala.so (gfx950 only)."""
    return os.path.join(_PKG, relative, 'lib', 'libpv_koala.so')


def developer_library_path(relative: str = '') -> str:
    """The -DKNS_DEV build of the same sources: kernel A/B selection, tuning knobs and the intermediate taps are steered
    by environment variables there (tests/ and tools/ only; the product library reads none of them)."""
    return os.path.join(_PKG, relative, 'lib', 'libpv_koala_dev.so')


def default_model_path(relative: str = '') -> str:
    """The default parameter file written by `build_native()`: the hand-built spectral gate with an adaptive noise floor
    (params.make_adaptive_gate) -- nothing in it is derived from an audio file."""
    return os.path.join(_PKG, relative, 'lib', 'koala_params.kns')


def build_native(force: bool = False) -> str:
    """Compile koala_amd/lib/libpv_koala.so with hipcc for gfx950 (no GPU needed) and write the default model."""
    lib, dev = default_library_path(), developer_library_path()
    src_dir = os.path.join(_PKG, 'csrc')
    deps = [os.path.join(src_dir, n) for n in os.listdir(src_dir)]
    deps += [os.path.join(_PKG, 'Makefile')]
    inc = os.path.join(_PKG, '..', 'include')
    if os.path.isdir(inc):
        deps += [os.path.join(inc, n) for n in os.listdir(inc)]

    def stale():
        return any(not os.path.exists(l) or any(os.path.getmtime(d) > os.path.getmtime(l) for d in deps) for l in (lib, dev))

    if force or stale():
        # several ranks (torchrun) may get here at once: one builds, the others wait on the lock and find the result;
        # the Makefile links to a temporary name and renames, so a library is never observed half-written
        import fcntl
        os.makedirs(os.path.dirname(lib), exist_ok=True)
        with open(os.path.join(os.path.dirname(lib), '.build.lock'), 'w') as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if force or stale():
                    subprocess.check_call(['make', '-C', _PKG, '-s'] + (['-B'] if force else []) + ['-j4', 'lib/libpv_koala.so', 'lib/libpv_koala_dev.so'])
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    model = default_model_path()
    tag, kind = model + '.kind', 'adaptive-gate-v4'
    if not os.path.exists(model) or not os.path.exists(tag) or open(tag).read().strip() != kind:
        from . import params
        params.write_params(model, params.make_adaptive_gate())
        with open(tag, 'w') as f:
            f.write(kind + '\n')
    return lib


__all__ = ['default_library_path', 'developer_library_path', 'default_model_path', 'build_native']
