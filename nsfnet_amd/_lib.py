"""ctypes binding of include/nsfnet_pinn.h.  The HIP library IS the product path:
there is no CPU or torch fallback - loading fails loudly when it is missing."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# $NSFNET_PINN_LIB selects another build of the same ABI (kernel experiments: scripts/abl_build.py)
LIB_PATH = os.environ.get("NSFNET_PINN_LIB") or os.path.join(_HERE, "lib", "libnsfnet_pinn.so")

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float

# name -> (restype, argtypes); mirrors include/nsfnet_pinn.h one to one
SIGNATURES = {
    "pinn_last_error": (ctypes.c_char_p, []),
    "pinn_abi_version": (c_int, []),
    "pinn_net_create": (c_int, [c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "pinn_net_destroy": (c_int, [c_void_p]),
    "pinn_net_set_precision": (c_int, [c_void_p, c_int, c_int, c_int]),
    "pinn_net_num_params": (c_int64, [c_void_p]),
    "pinn_net_prep_floats": (c_int64, [c_void_p]),
    "pinn_net_prepare": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "pinn_plan_create": (c_int, [c_void_p, c_int64, c_int, ctypes.POINTER(c_void_p)]),
    "pinn_plan_destroy": (c_int, [c_void_p]),
    "pinn_plan_padded_points": (c_int64, [c_void_p]),
    "pinn_plan_workspace_bytes": (c_int64, [c_void_p, c_int]),
    "pinn_plan_kernel": (ctypes.c_char_p, [c_void_p, c_int]),
    "pinn_residual_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_float, c_float, c_float, c_float,
                                      c_int, c_void_p, c_void_p]),
    "pinn_residual_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p, ctypes.POINTER(c_float), c_float, c_float,
                                       c_void_p, c_void_p]),
    "pinn_residual_backward_phases": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                              c_void_p, c_void_p, ctypes.POINTER(c_float), c_float, c_float,
                                              c_void_p, c_int, c_void_p]),
    "pinn_value_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                   ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p), ctypes.POINTER(c_float),
                                   c_int, c_void_p, c_void_p]),
    "pinn_value_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "pinn_grad_reduce": (c_int, [c_void_p, c_int, ctypes.POINTER(c_void_p), ctypes.POINTER(c_void_p),
                                 c_void_p, c_int, c_void_p]),
    "pinn_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                               c_float, c_float, c_float, c_float, c_int64, c_void_p]),
    "pinn_adam_step_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64,
                                   c_float, c_float, c_float, c_float, c_void_p, c_void_p]),
}

_lib = None


class PinnLibraryError(RuntimeError):
    pass


def load():
    """Load libnsfnet_pinn.so (built by nsfnet_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PinnLibraryError(
            "HIP library %s is missing - build it with `python -m nsfnet_amd.build`; "
            "there is no CPU fallback for the PINN hot path" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the .so does not export it
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().pinn_last_error()
        raise PinnLibraryError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
