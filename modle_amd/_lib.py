"""Loader of the native library (modle_amd/libmodle_hip.so).

There is deliberately no fallback: when the shared library is missing the import of any compute
entry point fails loudly instead of silently running something else.
"""
import ctypes as C
import os

import numpy as np

from .params import CellResult, Config, LaunchInfo, Task

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, os.environ.get("MODLE_HIP_LIB", "libmodle_hip.so"))

# every symbol include/modle_hip.h declares
EXPORTS = [
    "modle_hip_config_default", "modle_hip_config_transform", "modle_hip_interval_hash",
    "modle_hip_prng_seed", "modle_hip_prng_jump", "modle_hip_compute_num_lefs",
    "modle_hip_compute_contacts_per_epoch", "modle_hip_matrix_shape", "modle_hip_make_tasks",
    "modle_hip_stp_active_from_occupancy", "modle_hip_occupancy_from_stp", "modle_hip_create",
    "modle_hip_destroy", "modle_hip_add_interval", "modle_hip_submit_tasks", "modle_hip_launch",
    "modle_hip_wait", "modle_hip_last_kernel_ms", "modle_hip_get_results",
    "modle_hip_interval_outputs", "modle_hip_copy_outputs", "modle_hip_reset",
    "modle_hip_simulate_interval", "modle_hip_test_phases", "modle_hip_sort_barriers",
    "modle_hip_cancel", "modle_hip_test_units", "modle_hip_interval_done",
    "modle_hip_enable_state_log", "modle_hip_get_state_log", "modle_hip_set_wait_timeout",
    "modle_hip_last_launch_info",
    "modle_hip_runtime_versions", "modle_hip_size_class",
]

u64p = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(dtype=np.uint8, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")

_lib = None


def _share_the_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm bundles its own libamdhip64.so (same soname as the
    system's) and loads it by path; libmodle_hip.so names the runtime by soname.  Whichever of the two
    initialises second in a process that holds both copies finds no device (seen on the MI355X
    boxes: `import torch` after this library -> "No HIP GPUs are available", this library after
    torch's own load-by-path -> "no HIP device available").  When torch is installed its copy is
    therefore loaded FIRST, globally, so that this library's dependency resolves to it by soname and
    a later `import torch` finds its runtime already in place.  No torch import happens here;
    MODLE_HIP_OWN_RUNTIME=1 keeps the system's runtime (processes that never load torch)."""
    if os.environ.get("MODLE_HIP_OWN_RUNTIME", "") not in ("", "0"):
        return
    try:
        with open("/proc/self/maps") as f:
            if "libamdhip64" in f.read():
                return  # a runtime is already loaded: the dynamic linker will reuse it by soname
    except OSError:
        pass
    import importlib.util

    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    bundled = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        try:
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
        except OSError:
            pass  # (the system's runtime then serves this library alone)


def _warn_on_runtime_mismatch(L):
    """The library was built against the system's ROCm headers; when torch's bundled runtime serves it
    (above), the two versions may differ.  Same major version: same code-object and ABI generation; a
    different one is reported (MODLE_HIP_OWN_RUNTIME=1 keeps the system's runtime)."""
    try:
        built, rt = C.c_int(0), C.c_int(0)
        L.modle_hip_runtime_versions.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_int)]
        if L.modle_hip_runtime_versions(C.byref(built), C.byref(rt)) == 0 and \
                rt.value // 10_000_000 != built.value // 10_000_000:
            import warnings

            warnings.warn(f"libmodle_hip.so was built with HIP {built.value} and runs on HIP runtime {rt.value}: "
                          "set MODLE_HIP_OWN_RUNTIME=1 to use the system's runtime", RuntimeWarning)
    except AttributeError:
        pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            f"{SO_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc, gfx950). modle_amd has no CPU fallback.")
    _share_the_hip_runtime_with_torch()
    L = C.CDLL(SO_PATH)
    _warn_on_runtime_mismatch(L)
    P = C.POINTER
    err = [C.c_char_p, C.c_size_t]
    L.modle_hip_config_default.argtypes = [P(Config)]
    L.modle_hip_config_default.restype = None
    L.modle_hip_config_transform.argtypes = [P(Config)] + err
    L.modle_hip_interval_hash.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                          C.c_uint64]
    L.modle_hip_interval_hash.restype = C.c_uint64
    L.modle_hip_prng_seed.argtypes = [C.c_uint64, P(C.c_uint64)]
    L.modle_hip_prng_seed.restype = None
    L.modle_hip_prng_jump.argtypes = [P(C.c_uint64)]
    L.modle_hip_prng_jump.restype = None
    L.modle_hip_compute_num_lefs.argtypes = [P(Config), C.c_uint64]
    L.modle_hip_compute_num_lefs.restype = C.c_uint64
    L.modle_hip_compute_contacts_per_epoch.argtypes = [P(Config), C.c_uint64]
    L.modle_hip_compute_contacts_per_epoch.restype = C.c_uint64
    L.modle_hip_matrix_shape.argtypes = [P(Config), C.c_uint64, P(C.c_uint64), P(C.c_uint64)]
    L.modle_hip_matrix_shape.restype = None
    L.modle_hip_make_tasks.argtypes = [P(Config), C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                       C.c_uint64, P(Task)]
    L.modle_hip_stp_active_from_occupancy.argtypes = [C.c_double, C.c_double]
    L.modle_hip_stp_active_from_occupancy.restype = C.c_double
    L.modle_hip_occupancy_from_stp.argtypes = [C.c_double, C.c_double]
    L.modle_hip_occupancy_from_stp.restype = C.c_double
    L.modle_hip_sort_barriers.argtypes = [u64p, u8p, f64p, f64p, C.c_size_t]
    L.modle_hip_sort_barriers.restype = None
    L.modle_hip_cancel.argtypes = [C.c_void_p] + err
    L.modle_hip_interval_done.argtypes = [C.c_void_p, C.c_int]
    L.modle_hip_enable_state_log.argtypes = [C.c_void_p, C.c_uint32] + err
    L.modle_hip_get_state_log.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_size_t,
                                          P(C.c_size_t)] + err
    L.modle_hip_create.argtypes = [P(Config), C.c_int] + err
    L.modle_hip_create.restype = C.c_void_p
    L.modle_hip_destroy.argtypes = [C.c_void_p]
    L.modle_hip_destroy.restype = None
    L.modle_hip_reset.argtypes = [C.c_void_p]
    L.modle_hip_add_interval.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, u64p, u8p, f64p,
                                         f64p, C.c_size_t, C.c_void_p, C.c_void_p] + err
    L.modle_hip_submit_tasks.argtypes = [C.c_void_p, C.c_int, P(Task), C.c_size_t] + err
    L.modle_hip_launch.argtypes = [C.c_void_p, C.c_void_p] + err
    L.modle_hip_wait.argtypes = [C.c_void_p] + err
    L.modle_hip_set_wait_timeout.argtypes = [C.c_void_p, C.c_double]
    L.modle_hip_last_launch_info.argtypes = [C.c_void_p, P(LaunchInfo)]
    L.modle_hip_last_kernel_ms.argtypes = [C.c_void_p, P(C.c_float)]
    L.modle_hip_get_results.argtypes = [C.c_void_p, C.c_int, P(CellResult), C.c_size_t]
    L.modle_hip_interval_outputs.argtypes = [C.c_void_p, C.c_int, P(C.c_void_p), P(C.c_void_p),
                                             P(C.c_uint64), P(C.c_uint64)]
    L.modle_hip_copy_outputs.argtypes = [C.c_void_p, C.c_int, C.c_void_p, P(C.c_uint64),
                                         C.c_void_p] + err
    L.modle_hip_simulate_interval.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, u64p, u8p,
                                              f64p, f64p, C.c_size_t, P(Task), C.c_size_t, u32p,
                                              C.c_uint64, C.c_uint64, P(C.c_uint64), C.c_void_p,
                                              P(CellResult)] + err
    L.modle_hip_test_phases.argtypes = ([C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64,
                                         C.c_size_t] + [u64p] * 9 +
                                        [C.c_size_t, u64p, u8p, u8p, P(C.c_uint64),
                                         P(C.c_uint64)] + err)
    L.modle_hip_test_units.argtypes = [C.c_void_p, C.c_uint32, u64p, C.c_size_t, C.c_uint64,
                                       C.c_uint64, C.c_void_p, P(C.c_uint64), u64p] + err
    for name in EXPORTS:
        getattr(L, name)  # raises AttributeError if a declared symbol is not exported
    _lib = L
    return L
