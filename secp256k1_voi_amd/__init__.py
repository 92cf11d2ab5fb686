"""secp256k1_voi_amd — MI355X (gfx950) batch engine for the verify / scalar-mult path of
Yawning/secp256k1-voi.

This module is the thin Python binding of the C-ABI in include/secp256k1_voi_amd.h
(ctypes; no torch types cross the boundary).  All compute happens in the HIP library
``libsecp256k1_voi_amd.so`` built from csrc/; there is NO CPU fallback: importing works
everywhere, but creating an :class:`Engine` raises if the library or a GPU is missing.

Encodings are the reference's canonical ones: 32-byte big-endian scalars / coordinates,
65-byte point records (0x04‖X‖Y, or 65 zero bytes for the identity).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

# More than four HIP streams in a process (this engine holds up to four; torch and RCCL bring theirs) share the
# runtime's four default hardware queues and then run one after the other where the engine means them to overlap
# (measured: +0.28 ms on a 5.0 ms step once a process group exists).  The runtime reads this when
# it initialises, i.e. at the first HIP call of the process: importing this package before that is enough.
if "GPU_MAX_HW_QUEUES" not in os.environ:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"      # (process-wide: export your own value to overrule it)
    import sys as _sys
    _t = _sys.modules.get("torch")
    if _t is not None and getattr(_t, "cuda", None) is not None and _t.cuda.is_initialized():
        import warnings
        warnings.warn("secp256k1_voi_amd: the HIP runtime was initialised before this package was imported, so "
                      "GPU_MAX_HW_QUEUES=8 comes too late; streams of the engine may share hardware queues "
                      "(export GPU_MAX_HW_QUEUES=8 in the environment instead)", RuntimeWarning)

_HERE = os.path.dirname(os.path.abspath(__file__))
# S2K_LIB: load another build of the library (a compile-time variant made with build(variant=...)), for
# same-box A/B runs that must not spend GPU time compiling
LIB_PATH = os.environ.get("S2K_LIB") or os.path.join(_HERE, "libsecp256k1_voi_amd.so")
CSRC = os.path.join(_HERE, "csrc")

REJECT_MALLEABLE = 1
BIP0066 = 2
FORCE_COMPLETE = 0x80000000
FORCE_WORKLIST = 0x40000000
ENCODING_ASN1, ENCODING_COMPACT, ENCODING_COMPACT_RECOVERABLE = 0, 1, 2

OP_MUL, OP_SQR, OP_ADD, OP_SUB, OP_NEG, OP_INV, OP_SQRT = range(7)
IMPL_COMPLETE, IMPL_FAST = 0, 1
CTX_WAIT_TABLES = 1
KEYS_OFF, KEYS_AUTO, KEYS_ALWAYS, KEYS_ADAPTIVE = 0, 1, 2, 3     # s2k_ctx_set_key_grouping (a new context: KEYS_ADAPTIVE)
KEYSET_AUTO, KEYSET_CHUNKS, KEYSET_JOINT, KEYSET_JOINT5, KEYSET_JOINT6 = 0, 1, 2, 3, 4   # s2k_keyset_create_ex
(HP_MUL, HP_SQR, HP_MUL_PLUS, HP_SQR_PLUS, HP_MUL_ADD_MUL, HP_MUL_ADD_SQR, HP_ADD, HP_NEGATE, HP_HALF, HP_NORMALIZE,
 HP_COND_NEGATE1, HP_INV, HP_SQRT, HP_EQ, HP_MUL_SMALL21, HP_NORMALIZE_WEAK, HP_JDBL, HP_JADD, HP_PT29_DBL, HP_PT29_ADD,
 HP_PT29_ADD_MIXED, HP_INV_GCD, HP_JADD_FULL, HP_PT29Q_DBL, HP_PT29Q_ADD, HP_XYZZ_ADD, HP_XYZZ_ROUND,
 HP_FER_MUL, HP_FER_MUL_PLUS, HP_FER_MUL_ADD_MUL, HP_FER_SMALL, HP_PT29R_DBL, HP_PT29R_ADD, HP_FER_SWAPS) = range(34)

IDENTITY = bytes(65)


def pinned_array(shape, dtype=np.uint8) -> np.ndarray:
    """A numpy array in page-locked host memory (s2k_host_alloc): host-buffer calls copy from it asynchronously.
    The memory is released when the array (and every view of it) is gone."""
    import weakref
    lib = load_library()
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = lib.s2k_host_alloc(max(nbytes, 1))
    if not p:
        raise EngineError("s2k_host_alloc failed")
    buf = (C.c_uint8 * max(nbytes, 1)).from_address(p)
    weakref.finalize(buf, lib.s2k_host_free, p)
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def page_aligned_array(shape, dtype=np.uint8) -> np.ndarray:
    """A numpy array that owns whole pages (anonymous mmap, length rounded up to the page size): the kind of buffer to
    hand to host_register.  Registering a piece of the C heap pins and later unmaps pages that the allocator goes on
    using for other blocks (see s2k_host_register in the header)."""
    import mmap
    nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
    page = mmap.PAGESIZE
    m = mmap.mmap(-1, max((nbytes + page - 1) // page * page, page))
    return np.frombuffer(m, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def host_register(a: np.ndarray) -> None:
    """Pin an existing (contiguous) array for asynchronous copies (s2k_host_register); undo with host_unregister
    before the array is released.  The array must own whole pages (page_aligned_array); a heap array is refused."""
    import mmap
    lib = load_library()
    nbytes, base = a.nbytes, a
    while getattr(base, "base", None) is not None:
        base = base.base
    if isinstance(base, memoryview):
        base = base.obj
    if isinstance(base, mmap.mmap) and a.ctypes.data % mmap.PAGESIZE == 0:
        nbytes = (nbytes + mmap.PAGESIZE - 1) // mmap.PAGESIZE * mmap.PAGESIZE   # the mapping owns the rest of its last page
    if lib.s2k_host_register(a.ctypes.data, nbytes) != 0:
        raise EngineError(lib.s2k_last_error(None).decode())


def host_unregister(a: np.ndarray) -> None:
    lib = load_library()
    if lib.s2k_host_unregister(a.ctypes.data) != 0:
        raise EngineError(lib.s2k_last_error(None).decode())


class EngineError(RuntimeError):
    pass


def build_variant(name: str, extra_flags: str, verbose: bool = False) -> str:
    """Build libsecp256k1_voi_amd.<name>.so with `extra_flags` next to the default library (own object
    directory); select it at run time with S2K_LIB=<path>."""
    from concurrent.futures import ThreadPoolExecutor
    units = [f for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp"))]
    objdir = os.path.join(_HERE, "build", "variant_" + name)
    os.makedirs(objdir, exist_ok=True)
    out = os.path.join(_HERE, f"libsecp256k1_voi_amd.{name}.so")
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", '-DS2K_BUILD_FLAGS="%s"' % extra_flags.replace('"', "'")]
    flags += extra_flags.split()

    def compile_one(u):
        obj = os.path.join(objdir, os.path.splitext(u)[0] + ".o")
        subprocess.check_call(["hipcc", *flags, "-c", os.path.join(CSRC, u), "-o", obj])
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(units))) as ex:
        objs = list(ex.map(compile_one, units))
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out])
    if verbose:
        print("built", out)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP library for gfx950 with hipcc (cross-compiles without a GPU): one object
    per translation unit (in parallel), then one shared library.  The variant flags
    (S2K_EXTRA_FLAGS, e.g. "-DS2K_GT_BITS=16") are recorded next to the objects and compiled into
    s2k_build_config(); a library built with other flags than the ones asked for now is rebuilt,
    so an A/B script cannot leave a non-default kernel behind unnoticed."""
    from concurrent.futures import ThreadPoolExecutor
    units = [f for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".cpp"))]
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".cpp"))]
    deps.append(os.path.join(os.path.dirname(_HERE), "include", "secp256k1_voi_amd.h"))
    if os.environ.get("S2K_LIB"):      # a prebuilt variant was selected: nothing to build
        return LIB_PATH
    objdir = os.path.join(_HERE, "build")
    stamp = os.path.join(objdir, "flags.stamp")
    extra = " ".join(os.environ.get("S2K_EXTRA_FLAGS", "").split())
    try:
        with open(stamp) as f:
            built_with = f.read()
    except OSError:
        built_with = None
    fresh = os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in deps)
    if not force and fresh and (built_with == extra or (built_with is None and not extra and not os.path.isdir(objdir))):
        return LIB_PATH
    os.makedirs(objdir, exist_ok=True)
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", '-DS2K_BUILD_FLAGS="%s"' % extra.replace('"', "'")]
    flags += extra.split()
    # the stamp names the flags of the objects in the directory: it is withdrawn while they are being replaced, so that an
    # interrupted build (a variant build, say) cannot leave objects of other flags under a stamp that vouches for them
    if not (built_with == extra):
        force = True
    try:
        os.remove(stamp)
    except OSError:
        pass

    # an object is reused when it is newer than its own source and every header, and was built with these flags
    headers = [d for d in deps if d.endswith(".h")]
    newest_header = max(os.path.getmtime(h) for h in headers)
    same_flags = built_with == extra and built_with is not None

    def compile_one(u):
        obj = os.path.join(objdir, os.path.splitext(u)[0] + ".o")
        src = os.path.join(CSRC, u)
        if not force and same_flags and os.path.exists(obj) and os.path.getmtime(obj) >= max(newest_header, os.path.getmtime(src)):
            return obj
        cmd = ["hipcc", *flags, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(units))) as ex:
        objs = list(ex.map(compile_one, units))
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(extra)
    return LIB_PATH


_lib = None


def load_library() -> C.CDLL:
    """dlopen the engine.  Raises EngineError (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, sz, u32, ci = C.c_void_p, C.c_size_t, C.c_uint32, C.c_int
    lib.s2k_ctx_create.argtypes = [ci, C.POINTER(vp)]
    lib.s2k_ctx_destroy.argtypes = [vp]
    lib.s2k_ctx_destroy.restype = None
    lib.s2k_last_error.argtypes = [vp]
    lib.s2k_last_error.restype = C.c_char_p
    lib.s2k_version.restype = C.c_char_p
    lib.s2k_build_config.restype = C.c_char_p
    lib.s2k_ctx_profile.argtypes = [vp, ci]
    lib.s2k_ctx_profile_read.argtypes = [vp, vp, vp, sz, vp, vp]
    lib.s2k_ctx_profile_read_stages.argtypes = [vp, vp, vp, sz, vp, vp]
    lib.s2k_ctx_profile_msm.argtypes = [vp, ci]
    lib.s2k_ctx_profile_read_msm.argtypes = [vp, vp, vp]
    lib.s2k_ctx_set_key_grouping.argtypes = [vp, ci, u32, u32, u32]
    lib.s2k_ctx_key_grouping_stats.argtypes = [vp, vp]
    lib.s2k_ctx_key_grouping_adaptive.argtypes = [vp, vp, ci]
    lib.s2k_ecdsa_verify_batch.argtypes = [vp, sz, vp, vp, vp, vp, u32, vp]
    lib.s2k_ecdsa_verify_batch_device.argtypes = [vp, sz, vp, vp, vp, vp, u32, vp, vp]
    lib.s2k_ecdsa_recover_batch.argtypes = [vp, sz, vp, vp, vp, vp, u32, vp, vp]
    lib.s2k_ecdsa_recover_batch_device.argtypes = [vp, sz, vp, vp, vp, vp, u32, vp, vp, vp]
    lib.s2k_pack_valid_device.argtypes = [vp, sz, vp, vp, vp, vp]
    lib.s2k_keyset_create.argtypes = [vp, sz, vp, C.POINTER(vp)]
    lib.s2k_keyset_create_ex.argtypes = [vp, sz, vp, ci, C.POINTER(vp)]
    lib.s2k_keyset_layout.argtypes = [vp]
    lib.s2k_keyset_destroy.argtypes = [vp]
    lib.s2k_keyset_destroy.restype = None
    lib.s2k_keyset_size.argtypes = [vp]
    lib.s2k_keyset_size.restype = sz
    lib.s2k_keyset_device_bytes.argtypes = [vp]
    lib.s2k_keyset_device_bytes.restype = sz
    lib.s2k_keyset_valid_keys.argtypes = [vp, vp]
    lib.s2k_ecdsa_verify_batch_keyset.argtypes = [vp, vp, sz, vp, vp, vp, vp, u32, vp]
    lib.s2k_ecdsa_verify_batch_keyset_device.argtypes = [vp, vp, sz, vp, vp, vp, vp, u32, vp, vp]
    lib.s2k_ecdsa_verify_batch_keyset_submit.argtypes = [vp, vp, sz, vp, vp, vp, vp, u32, vp, vp]
    lib.s2k_host_alloc.argtypes = [sz]
    lib.s2k_host_alloc.restype = vp
    lib.s2k_host_free.argtypes = [vp]
    lib.s2k_host_free.restype = None
    lib.s2k_host_register.argtypes = [vp, sz]
    lib.s2k_host_unregister.argtypes = [vp]
    lib.s2k_ecdsa_workspace_bytes.argtypes = [sz]
    lib.s2k_ecdsa_workspace_bytes.restype = sz
    lib.s2k_ctx_device_bytes.argtypes = [vp, sz]
    lib.s2k_ctx_device_bytes.restype = sz
    lib.s2k_parse_asn1_signature.argtypes = [C.c_char_p, sz, C.c_char_p, C.c_char_p]
    lib.s2k_parse_compact_signature.argtypes = [C.c_char_p, sz, C.c_char_p, C.c_char_p]
    lib.s2k_is_valid_signature_encoding_bip0066.argtypes = [C.c_char_p, sz]
    lib.s2k_ecdsa_verify_encoded_batch.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, ci, sz, u32, vp]
    u64 = C.c_uint64
    lib.s2k_ecdsa_verify_batch_submit.argtypes = [vp, sz, vp, vp, vp, vp, u32, vp, C.POINTER(u64)]
    lib.s2k_ecdsa_verify_encoded_batch_submit.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, ci, sz, u32, vp, C.POINTER(u64)]
    lib.s2k_wait.argtypes = [vp, u64]
    lib.s2k_poll.argtypes = [vp, u64]
    lib.s2k_wait_all.argtypes = [vp]
    lib.s2k_device_count.restype = ci
    lib.s2k_group_create.argtypes = [C.POINTER(ci), sz, C.POINTER(vp)]
    lib.s2k_group_create_ex.argtypes = [C.POINTER(ci), sz, ci, u32, C.POINTER(vp)]
    lib.s2k_debug_gt_swap_in_call.argtypes = [vp, ci]
    lib.s2k_group_destroy.argtypes = [vp]
    lib.s2k_group_destroy.restype = None
    lib.s2k_group_size.argtypes = [vp]
    lib.s2k_group_size.restype = sz
    lib.s2k_group_last_error.argtypes = [vp]
    lib.s2k_group_last_error.restype = C.c_char_p
    lib.s2k_group_set_key_grouping.argtypes = [vp, ci, u32, u32, u32]
    lib.s2k_group_set_small_batch_max.argtypes = [vp, u32]
    lib.s2k_group_set_mid_batch_max.argtypes = [vp, u32]
    lib.s2k_group_schnorr_batch_verify_rlc.argtypes = [vp, sz, vp, vp, vp, sz, vp, vp, vp]
    lib.s2k_group_multi_scalar_mult.argtypes = [vp, sz, vp, vp, vp]
    lib.s2k_group_keyset_create.argtypes = [vp, sz, vp, ci, vp]
    lib.s2k_group_keyset_destroy.argtypes = [vp]
    lib.s2k_group_keyset_destroy.restype = None
    lib.s2k_group_keyset_size.argtypes = [vp]
    lib.s2k_group_keyset_size.restype = sz
    lib.s2k_group_keyset_layout.argtypes = [vp]
    lib.s2k_group_keyset_device_bytes.argtypes = [vp]
    lib.s2k_group_keyset_device_bytes.restype = sz
    lib.s2k_group_ecdsa_verify_batch_keyset.argtypes = [vp, vp, sz, vp, vp, vp, vp, u32, vp]
    lib.s2k_group_ecdsa_verify_batch_keyset_submit.argtypes = [vp, vp, sz, vp, vp, vp, vp, u32, vp, vp]
    lib.s2k_group_ecdsa_verify_batch.argtypes = [vp, sz, vp, vp, vp, vp, u32, vp]
    lib.s2k_group_ecdsa_verify_batch_submit.argtypes = [vp, sz, vp, vp, vp, vp, u32, vp, C.POINTER(u64)]
    lib.s2k_group_wait.argtypes = [vp, u64]
    lib.s2k_group_ecdsa_verify_encoded_batch.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, ci, sz, u32, vp]
    lib.s2k_group_ecdsa_verify_encoded_batch_submit.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, ci, sz, u32, vp, C.POINTER(u64)]
    lib.s2k_group_member_stats.argtypes = [vp, vp]
    lib.s2k_schnorr_verify_batch.argtypes = [vp, sz, vp, vp, vp, sz, vp, u32, vp]
    lib.s2k_schnorr_verify_batch_device.argtypes = [vp, sz, vp, vp, vp, sz, vp, u32, vp, vp]
    lib.s2k_schnorr_verify_batch_keyset.argtypes = [vp, vp, sz, vp, vp, vp, sz, vp, u32, vp]
    lib.s2k_schnorr_verify_batch_keyset_device.argtypes = [vp, vp, sz, vp, vp, vp, sz, vp, u32, vp, vp]
    lib.s2k_schnorr_verify_batch_keyset_submit.argtypes = [vp, vp, sz, vp, vp, vp, sz, vp, u32, vp, vp]
    lib.s2k_schnorr_batch_verify_rlc.argtypes = [vp, sz, vp, vp, vp, sz, vp, vp, C.POINTER(ci)]
    lib.s2k_schnorr_batch_verify_rlc_device.argtypes = [vp, sz, vp, vp, vp, sz, vp, vp, C.POINTER(ci), vp]
    lib.s2k_schnorr_verify_batch_bisect.argtypes = [vp, sz, vp, vp, vp, sz, vp, vp, vp, vp]
    lib.s2k_schnorr_verify_batch_bisect_device.argtypes = [vp, sz, vp, vp, vp, sz, vp, vp, vp, vp, vp]
    lib.s2k_scalar_base_mult_batch.argtypes = [vp, sz, vp, vp]
    lib.s2k_scalar_mult_batch.argtypes = [vp, sz, vp, vp, vp]
    lib.s2k_double_scalar_mult_basepoint_batch.argtypes = [vp, sz, vp, vp, vp, vp]
    lib.s2k_double_scalar_mult_basepoint_batch_ex.argtypes = [vp, u32, sz, vp, vp, vp, vp]
    lib.s2k_fp_op_batch_ex.argtypes = [vp, u32, ci, u32, sz, vp, vp, vp, vp]
    lib.s2k_fn_split_glv_batch_ex.argtypes = [vp, u32, sz, vp, vp, vp, vp]
    lib.s2k_point_add_batch.argtypes = [vp, sz, vp, vp, vp]
    lib.s2k_point_double_batch.argtypes = [vp, sz, vp, vp]
    lib.s2k_point_decode_batch.argtypes = [vp, sz, sz, vp, vp, vp]
    lib.s2k_multi_scalar_mult.argtypes = [vp, sz, vp, vp, vp]
    lib.s2k_multi_scalar_mult_device.argtypes = [vp, sz, vp, vp, vp, vp]
    lib.s2k_fp_op_batch.argtypes = [vp, ci, sz, vp, vp, vp, vp]
    lib.s2k_fn_op_batch.argtypes = [vp, ci, sz, vp, vp, vp, vp]
    lib.s2k_fn_split_glv_batch.argtypes = [vp, sz, vp, vp, vp]
    lib.s2k_debug_gtable_entry.argtypes = [vp, C.c_uint, C.c_uint, vp]
    lib.s2k_ct_scalar_mult.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    lib.s2k_ct_scalar_base_mult.argtypes = [C.c_char_p, C.c_char_p]
    lib.s2k_ct_ecdh.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    lib.s2k_ct_debug_fe_mul_count.restype = C.c_uint64
    cp, pu64 = C.c_char_p, C.POINTER(C.c_uint64)
    lib.s2k_ct_point_add.argtypes = [cp, cp, cp]
    lib.s2k_ct_point_double.argtypes = [cp, cp]
    lib.s2k_ct_point_subtract.argtypes = [cp, cp, cp]
    lib.s2k_ct_point_negate.argtypes = [cp, cp]
    lib.s2k_ct_point_conditional_negate.argtypes = [cp, C.c_uint64, cp]
    lib.s2k_ct_point_conditional_select.argtypes = [cp, cp, C.c_uint64, cp]
    lib.s2k_ct_point_equal.argtypes = [cp, cp, pu64]
    lib.s2k_ct_point_is_identity.argtypes = [cp, pu64]
    lib.s2k_ct_point_is_y_odd.argtypes = [cp, pu64]
    lib.s2k_ct_scalar_op.argtypes = [ci, cp, cp, cp]
    lib.s2k_ct_scalar_conditional_select.argtypes = [cp, cp, C.c_uint64, cp]
    lib.s2k_ct_scalar_conditional_negate.argtypes = [cp, C.c_uint64, cp]
    lib.s2k_ct_scalar_predicate.argtypes = [ci, cp, cp, pu64]
    lib.s2k_ct_scalar_set_bytes.argtypes = [cp, cp, pu64]
    lib.s2k_ct_fe_op.argtypes = [ci, cp, cp, cp, pu64]
    lib.s2k_ct_multi_scalar_mult.argtypes = [sz, C.c_char_p, C.c_char_p, C.c_char_p]
    lib.s2k_device_pci_bus_id.argtypes = [ci, C.c_char_p, sz]
    lib.s2k_device_numa_node.argtypes = [ci]
    lib.s2k_bind_thread_to_node.argtypes = [ci]
    lib.s2k_topology_prefer_node.argtypes = [vp, sz, ci]
    lib.s2k_topology_node_count.argtypes = [C.c_char_p]
    lib.s2k_topology_numa_node_of_pci.argtypes = [C.c_char_p, C.c_char_p]
    lib.s2k_topology_node_cpus.argtypes = [C.c_char_p, ci, C.POINTER(ci), sz]
    lib.s2k_ctx_ticket_timing.argtypes = [vp, ci]
    lib.s2k_ticket_times.argtypes = [vp, C.c_uint64, C.POINTER(C.c_double)]
    lib.s2k_group_member_stats_ex.argtypes = [vp, vp]
    lib.s2k_group_gt_wait.argtypes = [vp]
    lib.s2k_group_shard_size.argtypes = [vp, sz]
    lib.s2k_group_shard_size.restype = sz
    lib.s2k_group_host_alloc.argtypes = [vp, sz, sz]
    lib.s2k_group_host_alloc.restype = vp
    lib.s2k_group_host_free.argtypes = [vp, vp]
    lib.s2k_group_host_free.restype = None
    lib.s2k_ctx_set_small_batch_max.argtypes = [vp, u32]
    lib.s2k_ctx_set_mid_batch_max.argtypes = [vp, u32]
    lib.s2k_ctx_create_ex.argtypes = [ci, ci, u32, C.POINTER(vp)]
    lib.s2k_set_generator_table_budget.argtypes = [sz]
    lib.s2k_set_generator_table_budget.restype = None
    lib.s2k_set_table_memory_budgets.argtypes = [sz, sz]
    lib.s2k_set_table_memory_budgets.restype = None
    lib.s2k_ctx_gt_info.argtypes = [vp, C.POINTER(C.c_uint64)]
    lib.s2k_ctx_gt_note.argtypes = [vp]
    lib.s2k_ctx_gt_note.restype = C.c_char_p
    lib.s2k_ctx_gt_wait.argtypes = [vp]
    lib.s2k_ct_ecdsa_sign_raw.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint8)]
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "s2k_ctx_create", "s2k_ctx_destroy", "s2k_last_error", "s2k_version", "s2k_build_config",
    "s2k_ctx_profile", "s2k_ctx_profile_read", "s2k_ctx_profile_read_stages", "s2k_ctx_profile_msm", "s2k_ctx_profile_read_msm",
    "s2k_ctx_set_key_grouping", "s2k_ctx_key_grouping_stats", "s2k_ctx_key_grouping_adaptive",
    "s2k_ecdsa_verify_batch", "s2k_ecdsa_verify_batch_device", "s2k_ecdsa_workspace_bytes", "s2k_ctx_device_bytes",
    "s2k_keyset_create", "s2k_keyset_create_ex", "s2k_keyset_layout", "s2k_keyset_destroy", "s2k_keyset_size", "s2k_keyset_device_bytes", "s2k_keyset_valid_keys",
    "s2k_ecdsa_verify_batch_keyset", "s2k_ecdsa_verify_batch_keyset_device", "s2k_ecdsa_verify_batch_keyset_submit",
    "s2k_pack_valid_device", "s2k_host_alloc", "s2k_host_free", "s2k_host_register", "s2k_host_unregister", "s2k_ecdsa_recover_batch", "s2k_ecdsa_recover_batch_device",
    "s2k_parse_asn1_signature", "s2k_parse_compact_signature", "s2k_is_valid_signature_encoding_bip0066",
    "s2k_ecdsa_verify_encoded_batch",
    "s2k_ecdsa_verify_batch_submit", "s2k_ecdsa_verify_encoded_batch_submit", "s2k_wait", "s2k_poll", "s2k_wait_all",
    "s2k_device_count", "s2k_group_create", "s2k_group_destroy", "s2k_group_size", "s2k_group_last_error",
    "s2k_group_set_key_grouping", "s2k_group_set_small_batch_max", "s2k_group_set_mid_batch_max", "s2k_group_ecdsa_verify_batch", "s2k_group_ecdsa_verify_batch_submit", "s2k_group_wait",
    "s2k_group_ecdsa_verify_encoded_batch", "s2k_group_ecdsa_verify_encoded_batch_submit",
    "s2k_group_member_stats", "s2k_group_schnorr_batch_verify_rlc", "s2k_group_multi_scalar_mult",
    "s2k_group_keyset_create", "s2k_group_keyset_destroy", "s2k_group_keyset_size", "s2k_group_keyset_layout", "s2k_group_keyset_device_bytes",
    "s2k_group_ecdsa_verify_batch_keyset", "s2k_group_ecdsa_verify_batch_keyset_submit",
    "s2k_schnorr_verify_batch", "s2k_schnorr_verify_batch_device",
    "s2k_schnorr_verify_batch_keyset", "s2k_schnorr_verify_batch_keyset_device", "s2k_schnorr_verify_batch_keyset_submit",
    "s2k_schnorr_batch_verify_rlc", "s2k_schnorr_batch_verify_rlc_device",
    "s2k_schnorr_verify_batch_bisect", "s2k_schnorr_verify_batch_bisect_device",
    "s2k_scalar_base_mult_batch", "s2k_scalar_mult_batch", "s2k_double_scalar_mult_basepoint_batch",
    "s2k_point_add_batch", "s2k_point_double_batch", "s2k_point_decode_batch",
    "s2k_multi_scalar_mult", "s2k_multi_scalar_mult_device",
    "s2k_fp_op_batch", "s2k_fn_op_batch", "s2k_fn_split_glv_batch", "s2k_debug_gtable_entry", "s2k_generator_window_bits",
    "s2k_double_scalar_mult_basepoint_batch_ex", "s2k_fp_op_batch_ex", "s2k_fn_split_glv_batch_ex",
    "s2k_ct_scalar_mult", "s2k_ct_scalar_base_mult", "s2k_ct_ecdh", "s2k_ct_ecdsa_sign_raw", "s2k_ct_debug_fe_mul_count",
    "s2k_ct_multi_scalar_mult",
    "s2k_device_pci_bus_id", "s2k_device_numa_node", "s2k_bind_thread_to_node", "s2k_topology_prefer_node", "s2k_topology_node_count",
    "s2k_topology_numa_node_of_pci", "s2k_topology_node_cpus", "s2k_ctx_ticket_timing", "s2k_ticket_times",
    "s2k_group_member_stats_ex", "s2k_group_gt_wait", "s2k_group_shard_size", "s2k_group_host_alloc", "s2k_group_host_free",
    "s2k_ctx_set_small_batch_max", "s2k_ctx_set_mid_batch_max", "s2k_ctx_create_ex", "s2k_set_generator_table_budget", "s2k_set_table_memory_budgets", "s2k_ctx_gt_info", "s2k_ctx_gt_note", "s2k_ctx_gt_wait",
    "s2k_group_create_ex", "s2k_debug_gt_swap_in_call",
    "s2k_ct_point_add", "s2k_ct_point_double", "s2k_ct_point_subtract", "s2k_ct_point_negate", "s2k_ct_point_conditional_negate",
    "s2k_ct_point_conditional_select", "s2k_ct_point_equal", "s2k_ct_point_is_identity", "s2k_ct_point_is_y_odd",
    "s2k_ct_scalar_op", "s2k_ct_scalar_conditional_select", "s2k_ct_scalar_conditional_negate", "s2k_ct_scalar_predicate",
    "s2k_ct_scalar_set_bytes", "s2k_ct_fe_op",
]


def _arr(x, width, n=None):
    """bytes / list of bytes / ndarray -> contiguous uint8 array of shape (n, width)."""
    if isinstance(x, (list, tuple)):
        x = b"".join(x)
    if isinstance(x, (bytes, bytearray, memoryview)):
        a = np.frombuffer(bytes(x), dtype=np.uint8)
    else:
        a = np.ascontiguousarray(x, dtype=np.uint8).reshape(-1)
    if a.size % width:
        raise ValueError(f"buffer length {a.size} is not a multiple of {width}")
    a = a.reshape(-1, width)
    if n is not None and a.shape[0] != n:
        # the reference panics on length mismatch (point_mul_multi.go:27-29)
        raise ValueError(f"length mismatch: expected {n} items, got {a.shape[0]}")
    return np.ascontiguousarray(a)


# ---- host-side parsing (no GPU needed) --------------------------------------------------------
def parse_asn1_signature(der: bytes):
    """ParseASN1Signature (secec/s11n.go:83): (r, s) as 32-byte strings, or None."""
    r, s = C.create_string_buffer(32), C.create_string_buffer(32)
    return (r.raw, s.raw) if load_library().s2k_parse_asn1_signature(der, len(der), r, s) == 0 else None


def parse_compact_signature(sig: bytes):
    """ParseCompactSignature (secec/s11n.go:129)."""
    r, s = C.create_string_buffer(32), C.create_string_buffer(32)
    return (r.raw, s.raw) if load_library().s2k_parse_compact_signature(sig, len(sig), r, s) == 0 else None


def is_valid_signature_encoding_bip0066(sig: bytes) -> bool:
    """bitcoin.IsValidSignatureEncodingBIP0066 (secec/bitcoin/asn1_shitcoin.go:13)."""
    return bool(load_library().s2k_is_valid_signature_encoding_bip0066(sig, len(sig)))


# ---- constant-time twins on the host CPU (no GPU needed) -------------------------------------
def ct_scalar_mult(k: bytes, point65: bytes):
    """Point.ScalarMult (point_mul_glv.go:257), constant time, CPU.  Returns the 65-byte record or None
    for a malformed point."""
    out = C.create_string_buffer(65)
    return out.raw if load_library().s2k_ct_scalar_mult(bytes(k), bytes(point65), out) == 0 else None


def ct_multi_scalar_mult(scalars, points):
    """Point.MultiScalarMult (point_mul_multi.go:25-67), constant time in the scalars, CPU: Straus with masked table
    scans; one term is Point.ScalarMult, none is the identity.  `scalars`: 32-byte strings, `points`: 65-byte records.
    Returns the 65-byte record of the sum, or None for a malformed point.  A length mismatch raises (the reference
    panics, :27-29)."""
    scalars, points = [bytes(x) for x in scalars], [bytes(x) for x in points]
    if len(scalars) != len(points):
        raise ValueError("secp256k1: len(scalars) != len(points)")
    if any(len(x) != 32 for x in scalars) or any(len(x) != 65 for x in points):
        raise ValueError("scalars are 32 bytes, point records 65")
    out = C.create_string_buffer(65)
    rc = load_library().s2k_ct_multi_scalar_mult(len(scalars), b"".join(scalars) or None, b"".join(points) or None, out)
    if rc == -4:
        raise MemoryError("s2k_ct_multi_scalar_mult: out of memory")
    return out.raw if rc == 0 else None


# ---- single operations, constant time, host CPU (s2k_ct_point_* / s2k_ct_scalar_* / s2k_ct_fe_op): what the reference's
# Point / Scalar / field.Element methods bind to.  A malformed operand raises ValueError (the reference's constructors fail).
def _ct_out(fn, n, *args):
    out = C.create_string_buffer(n)
    rc = fn(*args, out)
    if rc != 0:
        raise ValueError(f"{fn.__name__}: operand the reference's type cannot hold ({rc})")
    return out.raw


def _ct_flag(fn, *args) -> int:
    v = C.c_uint64(7)
    rc = fn(*args, C.byref(v))
    if rc != 0:
        raise ValueError(f"{fn.__name__}: operand the reference's type cannot hold ({rc})")
    return int(v.value)


def ct_point_add(a65, b65): return _ct_out(load_library().s2k_ct_point_add, 65, bytes(a65), bytes(b65))
def ct_point_double(a65): return _ct_out(load_library().s2k_ct_point_double, 65, bytes(a65))
def ct_point_subtract(a65, b65): return _ct_out(load_library().s2k_ct_point_subtract, 65, bytes(a65), bytes(b65))
def ct_point_negate(a65): return _ct_out(load_library().s2k_ct_point_negate, 65, bytes(a65))
def ct_point_conditional_negate(a65, ctrl): return _ct_out(load_library().s2k_ct_point_conditional_negate, 65, bytes(a65), int(ctrl))
def ct_point_conditional_select(a65, b65, ctrl): return _ct_out(load_library().s2k_ct_point_conditional_select, 65, bytes(a65), bytes(b65), int(ctrl))
def ct_point_equal(a65, b65): return _ct_flag(load_library().s2k_ct_point_equal, bytes(a65), bytes(b65))
def ct_point_is_identity(a65): return _ct_flag(load_library().s2k_ct_point_is_identity, bytes(a65))
def ct_point_is_y_odd(a65): return _ct_flag(load_library().s2k_ct_point_is_y_odd, bytes(a65))
def ct_scalar_op(op, a32, b32=None): return _ct_out(load_library().s2k_ct_scalar_op, 32, int(op), bytes(a32), None if b32 is None else bytes(b32))
def ct_scalar_conditional_select(a32, b32, ctrl): return _ct_out(load_library().s2k_ct_scalar_conditional_select, 32, bytes(a32), bytes(b32), int(ctrl))
def ct_scalar_conditional_negate(a32, ctrl): return _ct_out(load_library().s2k_ct_scalar_conditional_negate, 32, bytes(a32), int(ctrl))
def ct_scalar_predicate(what, a32, b32=None): return _ct_flag(load_library().s2k_ct_scalar_predicate, int(what), bytes(a32), None if b32 is None else bytes(b32))


def ct_scalar_set_bytes(src32):
    """Scalar.SetBytes -> (src mod n, did_reduce)"""
    out, did = C.create_string_buffer(32), C.c_uint64(7)
    if load_library().s2k_ct_scalar_set_bytes(bytes(src32), out, C.byref(did)) != 0:
        raise ValueError("s2k_ct_scalar_set_bytes")
    return out.raw, int(did.value)


def ct_fe_op(op, a32, b32=None):
    """field.Element operation -> (out, flag); flag is 1 except for OP_SQRT of a non-square"""
    out, flag = C.create_string_buffer(32), C.c_uint64(7)
    rc = load_library().s2k_ct_fe_op(int(op), bytes(a32), None if b32 is None else bytes(b32), out, C.byref(flag))
    if rc != 0:
        raise ValueError(f"s2k_ct_fe_op: operand the reference's type cannot hold ({rc})")
    return out.raw, int(flag.value)


def ct_scalar_base_mult(k: bytes) -> bytes:
    """Point.ScalarBaseMult (point_mul_table.go:168), constant time, CPU."""
    out = C.create_string_buffer(65)
    rc = load_library().s2k_ct_scalar_base_mult(bytes(k), out)
    if rc != 0:
        raise EngineError(f"s2k_ct_scalar_base_mult failed ({rc})")
    return out.raw


def ct_ecdh(priv32: bytes, pub65: bytes):
    """PrivateKey.ECDH (secec/secec.go:53): the shared x-coordinate, or None for invalid inputs."""
    out = C.create_string_buffer(32)
    return out.raw if load_library().s2k_ct_ecdh(bytes(priv32), bytes(pub65), out) == 0 else None


def ct_ecdsa_sign_raw(priv32: bytes, digest32: bytes, nonce32: bytes):
    """(r, s, recovery_id) for a caller-supplied nonce (secec/ecdsa.go:335-390), or None when the
    nonce has to be redrawn."""
    r, s, rid = C.create_string_buffer(32), C.create_string_buffer(32), C.c_uint8(0)
    rc = load_library().s2k_ct_ecdsa_sign_raw(bytes(priv32), bytes(digest32), bytes(nonce32), r, s, C.byref(rid))
    return (r.raw, s.raw, int(rid.value)) if rc == 0 else None


def _concat(items):
    offs = np.zeros(len(items) + 1, dtype=np.uint64)
    if items:
        offs[1:] = np.cumsum([len(x) for x in items], dtype=np.uint64)
    blob = np.frombuffer(b"".join(items) or b"\0", dtype=np.uint8)
    return blob, offs


def _out_array(out, n):
    """The verdict array of a submit call: a fresh one, or the caller's - which the library writes n bytes into, later, from
    another thread: it has to be exactly a writable C-contiguous uint8 array of n items."""
    if out is None:
        return np.zeros(n, dtype=np.uint8)
    if not (isinstance(out, np.ndarray) and out.dtype == np.uint8 and out.shape == (n,) and out.flags["C_CONTIGUOUS"] and out.flags["WRITEABLE"]):
        raise ValueError(f"out must be a writable C-contiguous uint8 array of shape ({n},)")
    return out


class _TicketOwner:
    """Keeps the buffers of the batches in flight alive.  The library reads the inputs and writes the verdicts of a ticket
    asynchronously (DMA from pinned arrays; s2k_internal_pipe_retire copies verdicts on the fifth submit, in s2k_wait_all and
    in s2k_ctx_destroy): a Ticket that is dropped without wait() must not take its arrays with it.  A ticket
    leaves the table when it is waited for, or when the library has retired it: at most four are in flight, the submit of
    ticket t + 4 retires ticket t before it returns."""
    _IN_FLIGHT = 4

    def _hold(self, ticket: int, out, keep):
        if not hasattr(self, "_inflight"):
            self._inflight = {}
        self._inflight[ticket] = (out, keep)
        for t in [t for t in self._inflight if t + self._IN_FLIGHT <= ticket]:
            del self._inflight[t]
        return Ticket(self, ticket, out)

    def _release(self, ticket: int):
        getattr(self, "_inflight", {}).pop(ticket, None)


class Engine(_TicketOwner):
    """One context bound to one GPU (s2k_ctx)."""

    def __init__(self, device: int = 0, gt_bits: int = 0, wait_tables: bool = False):
        """gt_bits 0: automatic generator tables (usable at once on a narrow table, the wide one is built in the background);
        16..26: exactly that window width, built before the constructor returns.  wait_tables: return only when the
        background build has ended (s2k_ctx_create_ex)."""
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.s2k_ctx_create_ex(device, int(gt_bits), CTX_WAIT_TABLES if wait_tables else 0, C.byref(h))
        if rc != 0:
            raise EngineError(f"s2k_ctx_create failed ({rc}): {self._lib.s2k_last_error(None).decode()}")
        self._h = h
        self.device = device

    def gt_info(self) -> dict:
        """generator tables of this context's device: window bits in use now, bits the background build aims for, whether it
        is running, bytes held, and the reason for the choice (s2k_ctx_gt_info / s2k_ctx_gt_note)"""
        info = (C.c_uint64 * 4)()
        self._check(self._lib.s2k_ctx_gt_info(self._h, info))
        return {"bits": int(info[0]), "target_bits": int(info[1]), "building": bool(info[2]), "bytes": int(info[3]),
                "note": self._lib.s2k_ctx_gt_note(self._h).decode()}

    def debug_gt_swap_in_call(self, bits: int):
        """test hook (s2k_debug_gt_swap_in_call): the next ecdsa_verify_batch publishes the `bits`-wide generator table for the
        device's automatic contexts between its ladder launch and its worklist launch"""
        self._check(self._lib.s2k_debug_gt_swap_in_call(self._h, int(bits)))

    def set_small_batch_max(self, max_n: int):
        """batches of up to max_n signatures take the wave-per-signature ladder (s2k_ctx_set_small_batch_max; 0: never)"""
        self._check(self._lib.s2k_ctx_set_small_batch_max(self._h, int(max_n)))

    def set_mid_batch_max(self, max_n: int):
        """ECDSA batches above the small-batch threshold and up to max_n take the four-lanes-per-signature ladder (0: never)"""
        self._check(self._lib.s2k_ctx_set_mid_batch_max(self._h, int(max_n)))

    def gt_wait(self) -> int:
        """block until the background build of the wide generator tables has ended; the window bits in use then"""
        return int(self._lib.s2k_ctx_gt_wait(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.s2k_ctx_destroy(self._h)      # (retires the batches in flight: their buffers are needed until it returns)
            self._h = None
            self._inflight = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise EngineError(f"engine call failed ({rc}): {self._lib.s2k_last_error(self._h).decode()}")

    def _wait(self, ticket):
        self._check(self._lib.s2k_wait(self._h, ticket))

    def _poll(self, ticket) -> bool:
        rc = self._lib.s2k_poll(self._h, ticket)
        if rc == 1:          # S2K_PENDING
            return False
        self._check(rc)
        return True

    # ---- hot path -------------------------------------------------------------------
    def ecdsa_verify_batch(self, pub_xy, digest32, r, s, reject_malleable: bool = False,
                           force_complete: bool = False, force_worklist: bool = False) -> np.ndarray:
        """valid bits (uint8 0/1) for n (pubkey, digest, r, s) tuples; host buffers."""
        r = _arr(r, 32)
        n = r.shape[0]
        pub_xy, digest32, s = _arr(pub_xy, 64, n), _arr(digest32, 32, n), _arr(s, 32, n)
        out = np.zeros(n, dtype=np.uint8)
        self._check(self._lib.s2k_ecdsa_verify_batch(self._h, n, pub_xy.ctypes.data, digest32.ctypes.data,
                                                     r.ctypes.data, s.ctypes.data,
                                                     (REJECT_MALLEABLE if reject_malleable else 0) |
                                                     (FORCE_COMPLETE if force_complete else 0) |
                                                     (FORCE_WORKLIST if force_worklist else 0), out.ctypes.data))
        return out

    # ---- submit / wait -------------------------------------------------------------------
    def ecdsa_verify_batch_submit(self, pub_xy, digest32, r, s, out=None, reject_malleable: bool = False) -> "Ticket":
        """s2k_ecdsa_verify_batch_submit: enqueue the batch and return; `Ticket.wait()` gives the valid bits.  The input
        arrays are used as they are (no copies: pass contiguous uint8 arrays of the right shapes, page-locked ones
        from pinned_array for asynchronous transfers) and are kept alive by the ticket."""
        arrs = []
        n = None
        for a, w in ((pub_xy, 64), (digest32, 32), (r, 32), (s, 32)):
            if not (isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]):
                a = _arr(a, w)
            a = a.reshape(-1, w)
            if n is None:
                n = a.shape[0]
            elif a.shape[0] != n:
                raise ValueError(f"length mismatch: expected {n} items, got {a.shape[0]}")
            arrs.append(a)
        out = _out_array(out, n)
        t = C.c_uint64(0)
        self._check(self._lib.s2k_ecdsa_verify_batch_submit(self._h, n, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data,
                                                            arrs[3].ctypes.data, REJECT_MALLEABLE if reject_malleable else 0,
                                                            out.ctypes.data, C.byref(t)))
        return self._hold(int(t.value), out, arrs)

    def ecdsa_verify_encoded_batch_submit(self, pubs, digests, sigs, encoding=ENCODING_ASN1, digest_len=0,
                                          reject_malleable=False, bip0066=False) -> "Ticket":
        """s2k_ecdsa_verify_encoded_batch_submit for lists of byte strings (or pre-built (blob, offsets) pairs)."""
        def cat(x):
            return x if isinstance(x, tuple) else _concat(list(x))
        (pb, po), (db, do), (sb, so) = cat(pubs), cat(digests), cat(sigs)
        n = len(po) - 1
        if len(do) - 1 != n or len(so) - 1 != n:
            raise ValueError("length mismatch")
        out = np.zeros(n, dtype=np.uint8)
        flags = (REJECT_MALLEABLE if reject_malleable else 0) | (BIP0066 if bip0066 else 0)
        t = C.c_uint64(0)
        self._check(self._lib.s2k_ecdsa_verify_encoded_batch_submit(self._h, n, pb.ctypes.data, po.ctypes.data, db.ctypes.data,
                                                                    do.ctypes.data, sb.ctypes.data, so.ctypes.data, encoding,
                                                                    digest_len, flags, out.ctypes.data, C.byref(t)))
        return self._hold(int(t.value), out, [pb, po, db, do, sb, so])

    def wait_all(self):
        self._check(self._lib.s2k_wait_all(self._h))
        self._inflight = {}

    # ---- key sets ----------------------------------------------------------------------
    def keyset_create(self, pub_xy, layout: int = 0) -> "KeySet":
        """Per-key tables of a fixed list of public keys (n_keys x 64 bytes), built once (s2k_keyset_create[_ex]);
        layout: KEYSET_AUTO (0), KEYSET_CHUNKS (1), KEYSET_JOINT (2)."""
        return KeySet(self, pub_xy, layout)

    def ecdsa_verify_batch_keyset(self, keyset, key_index, digest32, r, s, reject_malleable: bool = False,
                                  force_worklist: bool = False) -> np.ndarray:
        """valid bits for n (key index into `keyset`, digest, r, s) tuples; host buffers."""
        r = _arr(r, 32)
        n = r.shape[0]
        digest32, s = _arr(digest32, 32, n), _arr(s, 32, n)
        ki = np.ascontiguousarray(key_index, dtype=np.uint32).reshape(-1)
        if ki.shape[0] != n:
            raise ValueError(f"length mismatch: expected {n} key indices, got {ki.shape[0]}")
        out = np.zeros(n, dtype=np.uint8)
        self._check(self._lib.s2k_ecdsa_verify_batch_keyset(self._h, keyset._k, n, ki.ctypes.data, digest32.ctypes.data,
                                                            r.ctypes.data, s.ctypes.data,
                                                            (REJECT_MALLEABLE if reject_malleable else 0) |
                                                            (FORCE_WORKLIST if force_worklist else 0), out.ctypes.data))
        return out

    def ecdsa_verify_batch_keyset_submit(self, keyset, key_index, digest32, r, s, out=None, reject_malleable: bool = False) -> "Ticket":
        """s2k_ecdsa_verify_batch_keyset_submit: as ecdsa_verify_batch_submit, the keys named by their index in `keyset`
        (contiguous arrays are used as they are - uint32 indices, uint8 rows - and kept alive by the ticket)."""
        ki = key_index if (isinstance(key_index, np.ndarray) and key_index.dtype == np.uint32 and key_index.flags["C_CONTIGUOUS"]) \
            else np.ascontiguousarray(key_index, dtype=np.uint32)
        ki = ki.reshape(-1)
        n = ki.shape[0]
        arrs = [ki]
        for a in (digest32, r, s):
            if not (isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]):
                a = _arr(a, 32)
            a = a.reshape(-1, 32)
            if a.shape[0] != n:
                raise ValueError(f"length mismatch: expected {n} items, got {a.shape[0]}")
            arrs.append(a)
        out = _out_array(out, n)
        t = C.c_uint64(0)
        self._check(self._lib.s2k_ecdsa_verify_batch_keyset_submit(self._h, keyset._k, n, arrs[0].ctypes.data, arrs[1].ctypes.data,
                                                                   arrs[2].ctypes.data, arrs[3].ctypes.data,
                                                                   REJECT_MALLEABLE if reject_malleable else 0, out.ctypes.data,
                                                                   C.byref(t)))
        return self._hold(int(t.value), out, arrs + [keyset])

    def ecdsa_verify_batch_keyset_device(self, keyset, n, d_key_index, d_digest32, d_r, d_s, d_valid, flags=0, stream=0):
        self._check(self._lib.s2k_ecdsa_verify_batch_keyset_device(self._h, keyset._k, int(n), d_key_index, d_digest32, d_r, d_s,
                                                                   int(flags), d_valid, stream))

    def ecdsa_verify_encoded_batch(self, pubs, digests, sigs, encoding=ENCODING_ASN1, digest_len=0,
                                   reject_malleable=False, bip0066=False, force_complete=False) -> np.ndarray:
        """PublicKey.Verify(digest, sig, opts) for lists of SEC1 keys, digests and encoded signatures."""
        n = len(pubs)
        if len(digests) != n or len(sigs) != n:
            raise ValueError("length mismatch")
        pb, po = _concat(list(pubs))
        db, do = _concat(list(digests))
        sb, so = _concat(list(sigs))
        out = np.zeros(n, dtype=np.uint8)
        flags = (REJECT_MALLEABLE if reject_malleable else 0) | (BIP0066 if bip0066 else 0) | \
                (FORCE_COMPLETE if force_complete else 0)
        self._check(self._lib.s2k_ecdsa_verify_encoded_batch(self._h, n, pb.ctypes.data, po.ctypes.data, db.ctypes.data,
                                                             do.ctypes.data, sb.ctypes.data, so.ctypes.data, encoding,
                                                             digest_len, flags, out.ctypes.data))
        return out

    def ecdsa_verify_batch_device(self, n, d_pub_xy, d_digest32, d_r, d_s, d_valid, flags=0, stream=0):
        """Device-pointer form: integer device addresses, enqueued on `stream` (a hipStream_t value)."""
        self._check(self._lib.s2k_ecdsa_verify_batch_device(self._h, int(n), d_pub_xy, d_digest32, d_r, d_s,
                                                            int(flags), d_valid, stream))

    def schnorr_verify_batch(self, pk32, msgs, sig64, force_complete: bool = False) -> np.ndarray:
        """BIP-340: valid bits for n (x-only key, message, 64-byte signature) triples.
        `msgs` is a list of byte strings (any lengths) or an (n, L) uint8 array."""
        pk32 = _arr(pk32, 32)
        n = pk32.shape[0]
        sig64 = _arr(sig64, 64, n)
        flags = FORCE_COMPLETE if force_complete else 0
        out = np.zeros(n, dtype=np.uint8)
        if isinstance(msgs, (list, tuple)):
            if len(msgs) != n:
                raise ValueError(f"length mismatch: expected {n} messages, got {len(msgs)}")
            offs = np.zeros(n + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([len(m) for m in msgs], dtype=np.uint64)
            blob = np.frombuffer(b"".join(msgs) or b"\0", dtype=np.uint8)
            self._check(self._lib.s2k_schnorr_verify_batch(self._h, n, pk32.ctypes.data, blob.ctypes.data,
                                                           offs.ctypes.data, 0, sig64.ctypes.data, flags, out.ctypes.data))
        else:
            m = np.ascontiguousarray(msgs, dtype=np.uint8).reshape(n, -1) if n else np.zeros((0, 0), np.uint8)
            self._check(self._lib.s2k_schnorr_verify_batch(self._h, n, pk32.ctypes.data, m.ctypes.data if m.size else None,
                                                           None, m.shape[1], sig64.ctypes.data, flags, out.ctypes.data))
        return out

    def schnorr_verify_batch_keyset(self, keyset, key_index, msgs, sig64) -> np.ndarray:
        """BIP-340 over a key set (s2k_schnorr_verify_batch_keyset): signature i is verified under the x coordinate of key
        key_index[i] of `keyset` (a set of X || Y keys; the parity of Y does not matter)."""
        ki = np.ascontiguousarray(key_index, dtype=np.uint32).reshape(-1)
        n = ki.shape[0]
        sig64 = _arr(sig64, 64, n)
        out = np.zeros(n, dtype=np.uint8)
        if isinstance(msgs, (list, tuple)):
            if len(msgs) != n:
                raise ValueError(f"length mismatch: expected {n} messages, got {len(msgs)}")
            offs = np.zeros(n + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([len(m) for m in msgs], dtype=np.uint64)
            blob = np.frombuffer(b"".join(msgs) or b"\0", dtype=np.uint8)
            self._check(self._lib.s2k_schnorr_verify_batch_keyset(self._h, keyset._k, n, ki.ctypes.data, blob.ctypes.data, offs.ctypes.data, 0,
                                                                  sig64.ctypes.data, 0, out.ctypes.data))
        else:
            m = np.ascontiguousarray(msgs, dtype=np.uint8).reshape(n, -1) if n else np.zeros((0, 0), np.uint8)
            self._check(self._lib.s2k_schnorr_verify_batch_keyset(self._h, keyset._k, n, ki.ctypes.data, m.ctypes.data if m.size else None, None,
                                                                  m.shape[1], sig64.ctypes.data, 0, out.ctypes.data))
        return out

    def schnorr_verify_batch_keyset_submit(self, keyset, key_index, msgs, sig64, out=None) -> "Ticket":
        """s2k_schnorr_verify_batch_keyset_submit; `msgs` an (n, L) uint8 array (fixed-length messages) or a list of byte strings."""
        ki = np.ascontiguousarray(key_index, dtype=np.uint32).reshape(-1)
        n = ki.shape[0]
        sig64 = sig64 if (isinstance(sig64, np.ndarray) and sig64.dtype == np.uint8 and sig64.flags["C_CONTIGUOUS"]) else _arr(sig64, 64, n)
        out = _out_array(out, n)
        t = C.c_uint64(0)
        if isinstance(msgs, (list, tuple)):
            offs = np.zeros(n + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([len(m) for m in msgs], dtype=np.uint64)
            blob = np.frombuffer(b"".join(msgs) or b"\0", dtype=np.uint8)
            keep = [ki, blob, offs, sig64, keyset]
            self._check(self._lib.s2k_schnorr_verify_batch_keyset_submit(self._h, keyset._k, n, ki.ctypes.data, blob.ctypes.data, offs.ctypes.data, 0,
                                                                         sig64.ctypes.data, 0, out.ctypes.data, C.byref(t)))
        else:
            m = msgs if (isinstance(msgs, np.ndarray) and msgs.dtype == np.uint8 and msgs.flags["C_CONTIGUOUS"]) else np.ascontiguousarray(msgs, dtype=np.uint8)
            m = m.reshape(n, -1) if n else np.zeros((0, 0), np.uint8)
            keep = [ki, m, sig64, keyset]
            self._check(self._lib.s2k_schnorr_verify_batch_keyset_submit(self._h, keyset._k, n, ki.ctypes.data, m.ctypes.data if m.size else None, None,
                                                                         m.shape[1], sig64.ctypes.data, 0, out.ctypes.data, C.byref(t)))
        return self._hold(int(t.value), out, keep)

    def schnorr_verify_batch_keyset_device(self, keyset, n, d_key_index, d_msgs, msg_len, d_sig, d_valid, stream=0):
        self._check(self._lib.s2k_schnorr_verify_batch_keyset_device(self._h, keyset._k, int(n), d_key_index, d_msgs, None, int(msg_len), d_sig, 0,
                                                                     d_valid, stream))

    def schnorr_batch_verify_rlc(self, pk32, msgs, sig64, seed32: bytes | None = None) -> bool:
        """True iff every (key, message, signature) triple verifies — one MSM of n + 2K + 2 terms
        (K distinct keys: the coefficients of a key's signatures are summed first)."""
        pk32 = _arr(pk32, 32)
        n = pk32.shape[0]
        sig64 = _arr(sig64, 64, n)
        seed = np.frombuffer(seed32 if seed32 is not None else os.urandom(32), dtype=np.uint8)
        if seed.size != 32:
            raise ValueError("seed32 must be 32 bytes")
        res = C.c_int(0)
        if isinstance(msgs, (list, tuple)):
            if len(msgs) != n:
                raise ValueError(f"length mismatch: expected {n} messages, got {len(msgs)}")
            offs = np.zeros(n + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([len(m) for m in msgs], dtype=np.uint64)
            blob = np.frombuffer(b"".join(msgs) or b"\0", dtype=np.uint8)
            self._check(self._lib.s2k_schnorr_batch_verify_rlc(self._h, n, pk32.ctypes.data, blob.ctypes.data,
                                                               offs.ctypes.data, 0, sig64.ctypes.data, seed.ctypes.data,
                                                               C.byref(res)))
        else:
            m = np.ascontiguousarray(msgs, dtype=np.uint8).reshape(n, -1) if n else np.zeros((0, 0), np.uint8)
            self._check(self._lib.s2k_schnorr_batch_verify_rlc(self._h, n, pk32.ctypes.data,
                                                               m.ctypes.data if m.size else None, None,
                                                               m.shape[1] if n else 0, sig64.ctypes.data,
                                                               seed.ctypes.data, C.byref(res)))
        return bool(res.value)

    def schnorr_verify_batch_auto(self, pk32, msgs, sig64, seed32: bytes | None = None, return_stats: bool = False):
        """Valid bits like schnorr_verify_batch, at the cost of the whole-batch check when (as usual)
        everything verifies: one random-linear-combination MSM first; if that rejects, the failing
        signatures are located by bisection on the kept terms (s2k_schnorr_verify_batch_bisect)."""
        pk32 = _arr(pk32, 32)
        n = pk32.shape[0]
        sig64 = _arr(sig64, 64, n)
        seed = np.frombuffer(seed32 if seed32 is not None else os.urandom(32), dtype=np.uint8)
        if seed.size != 32:
            raise ValueError("seed32 must be 32 bytes")
        out = np.zeros(n, dtype=np.uint8)
        stats = np.zeros(4, dtype=np.uint32)
        if isinstance(msgs, (list, tuple)):
            if len(msgs) != n:
                raise ValueError(f"length mismatch: expected {n} messages, got {len(msgs)}")
            offs = np.zeros(n + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([len(m) for m in msgs], dtype=np.uint64)
            blob = np.frombuffer(b"".join(msgs) or b"\0", dtype=np.uint8)
            self._check(self._lib.s2k_schnorr_verify_batch_bisect(self._h, n, pk32.ctypes.data, blob.ctypes.data, offs.ctypes.data,
                                                                  0, sig64.ctypes.data, seed.ctypes.data, out.ctypes.data,
                                                                  stats.ctypes.data))
        else:
            m = np.ascontiguousarray(msgs, dtype=np.uint8).reshape(n, -1) if n else np.zeros((0, 0), np.uint8)
            self._check(self._lib.s2k_schnorr_verify_batch_bisect(self._h, n, pk32.ctypes.data, m.ctypes.data if m.size else None,
                                                                  None, m.shape[1] if n else 0, sig64.ctypes.data,
                                                                  seed.ctypes.data, out.ctypes.data, stats.ctypes.data))
        if return_stats:
            return out, {"sub_combinations": int(stats[0]), "verified_one_by_one": int(stats[1]), "levels": int(stats[2]),
                         "abandoned": bool(stats[3])}
        return out

    def ecdsa_recover_batch(self, digest32, r, s, recovery_id, force_complete: bool = False):
        """RecoverPublicKey over a batch: returns (pub65 (n,65) uint8, ok (n,) uint8)."""
        r = _arr(r, 32)
        n = r.shape[0]
        digest32, s = _arr(digest32, 32, n), _arr(s, 32, n)
        rid = np.ascontiguousarray(recovery_id, dtype=np.uint8).reshape(-1)
        if rid.shape[0] != n:
            raise ValueError("length mismatch")
        pub, ok = np.zeros((n, 65), dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        self._check(self._lib.s2k_ecdsa_recover_batch(self._h, n, digest32.ctypes.data, r.ctypes.data, s.ctypes.data,
                                                      rid.ctypes.data, FORCE_COMPLETE if force_complete else 0,
                                                      pub.ctypes.data, ok.ctypes.data))
        return pub, ok

    def pack_valid_device(self, n, d_valid, d_bitmap, d_count, stream=0):
        """valid bytes -> bitmap + uint64 count, all device pointers."""
        self._check(self._lib.s2k_pack_valid_device(self._h, int(n), d_valid, d_bitmap, d_count, stream))

    def profile(self, enable: bool = True):
        """Per-kernel HIP-event timing of ecdsa_verify_batch_device calls (s2k_ctx_profile)."""
        self._check(self._lib.s2k_ctx_profile(self._h, 1 if enable else 0))

    def profile_read(self, cap: int = 1024):
        """-> dict(calls, prep_ms, fast_ms, fallback_ms (sums), fast_each (ms per call), shader_mhz = mean of
        the clock seen by the first wave and by a wave of the final round of the last ladder launch)."""
        sums = (C.c_double * 3)()
        each = (C.c_double * cap)()
        calls, mhz = C.c_size_t(0), (C.c_double * 2)()
        self._check(self._lib.s2k_ctx_profile_read(self._h, sums, each, cap, C.byref(calls), mhz))
        k = int(calls.value)
        both = [m for m in (mhz[0], mhz[1]) if m > 0]
        return {"calls": k, "prep_ms": sums[0], "fast_ms": sums[1], "fallback_ms": sums[2],
                "fast_each": [each[i] for i in range(min(k, cap))], "shader_mhz": sum(both) / len(both) if both else 0.0,
                "shader_mhz_first_wave": float(mhz[0]), "shader_mhz_last_round": float(mhz[1])}

    def profile_read_stages(self, cap: int = 1024):
        """-> dict(calls, prep_ms, group_ms (grouping by key + per-key tables), fast_ms (ladder over the per-key
        tables; the general ladder when grouping is off), left_ms (general ladder over the ungrouped rest),
        fallback_ms: sums; fast_each; shader_mhz...) (s2k_ctx_profile_read_stages)."""
        sums = (C.c_double * 5)()
        each = (C.c_double * cap)()
        calls, mhz = C.c_size_t(0), (C.c_double * 2)()
        self._check(self._lib.s2k_ctx_profile_read_stages(self._h, sums, each, cap, C.byref(calls), mhz))
        k = int(calls.value)
        both = [m for m in (mhz[0], mhz[1]) if m > 0]
        return {"calls": k, "prep_ms": sums[0], "group_ms": sums[1], "fast_ms": sums[2], "left_ms": sums[3],
                "fallback_ms": sums[4], "fast_each": [each[i] for i in range(min(k, cap))],
                "shader_mhz": sum(both) / len(both) if both else 0.0,
                "shader_mhz_first_wave": float(mhz[0]), "shader_mhz_last_round": float(mhz[1])}

    def profile_msm(self, enable: bool = True):
        """Stage timing of the multi-scalar path (s2k_ctx_profile_msm)."""
        self._check(self._lib.s2k_ctx_profile_msm(self._h, 1 if enable else 0))

    def profile_read_msm(self):
        """-> dict(calls, front_ms, sort_ms, bucket_pass_ms, reduce_ms, tail_ms): sums over the calls since the last read."""
        sums = (C.c_double * 5)()
        calls = C.c_size_t(0)
        self._check(self._lib.s2k_ctx_profile_read_msm(self._h, sums, C.byref(calls)))
        return {"calls": int(calls.value), "front_ms": sums[0], "sort_ms": sums[1], "bucket_pass_ms": sums[2],
                "reduce_ms": sums[3], "tail_ms": sums[4]}

    def set_key_grouping(self, mode: int = KEYS_AUTO, min_group: int = 0, hash_bits: int = 0, max_tables: int = 0):
        """How ecdsa_verify_batch[_device] treats signatures that share a public key (s2k_ctx_set_key_grouping):
        KEYS_OFF = every signature through the general kernel, KEYS_AUTO = keys with at least `min_group` (default 4, the
        measured break-even) signatures in the batch get a per-key table, KEYS_ALWAYS = every key does, KEYS_ADAPTIVE (what a
        new engine starts with) = KEYS_AUTO that stops looking for repeated keys after two large batches without any and looks
        again every sixteenth batch (key_grouping_adaptive)."""
        self._check(self._lib.s2k_ctx_set_key_grouping(self._h, int(mode), int(min_group), int(hash_bits), int(max_tables)))

    def key_grouping_stats(self):
        """Of the last ecdsa_verify_batch_device call -> dict(keyed, tables, general, complete)."""
        st = (C.c_uint32 * 4)()
        self._check(self._lib.s2k_ctx_key_grouping_stats(self._h, st))
        return {"keyed": int(st[0]), "tables": int(st[1]), "general": int(st[2]), "complete": int(st[3])}

    def key_grouping_adaptive(self, reset: bool = False):
        """State of KEYS_ADAPTIVE, without synchronising -> dict(miss_streak, skip_left, skipped, probes, observed)."""
        st = (C.c_uint32 * 5)()
        self._check(self._lib.s2k_ctx_key_grouping_adaptive(self._h, st, 1 if reset else 0))
        return {"miss_streak": int(st[0]), "skip_left": int(st[1]), "skipped": int(st[2]), "probes": int(st[3]),
                "observed": int(st[4])}

    def workspace_bytes(self, n):
        return self._lib.s2k_ecdsa_workspace_bytes(int(n))

    def device_bytes(self, n):
        """Device memory this engine holds after a verification call of n signatures with the current grouping settings
        (generator tables + per-signature workspace + grouping arrays and per-key table buffer)."""
        return self._lib.s2k_ctx_device_bytes(self._h, int(n))

    # ---- group ----------------------------------------------------------------------
    def _points_out(self, n):
        return np.zeros((n, 65), dtype=np.uint8)

    def scalar_base_mult_batch(self, k):
        k = _arr(k, 32)
        out = self._points_out(k.shape[0])
        self._check(self._lib.s2k_scalar_base_mult_batch(self._h, k.shape[0], k.ctypes.data, out.ctypes.data))
        return out

    def scalar_mult_batch(self, k, points):
        k = _arr(k, 32)
        n = k.shape[0]
        points = _arr(points, 65, n)
        out = self._points_out(n)
        self._check(self._lib.s2k_scalar_mult_batch(self._h, n, k.ctypes.data, points.ctypes.data, out.ctypes.data))
        return out

    def double_scalar_mult_basepoint_batch(self, u1, u2, points):
        u1 = _arr(u1, 32)
        n = u1.shape[0]
        u2, points = _arr(u2, 32, n), _arr(points, 65, n)
        out = self._points_out(n)
        self._check(self._lib.s2k_double_scalar_mult_basepoint_batch(self._h, n, u1.ctypes.data, u2.ctypes.data,
                                                                     points.ctypes.data, out.ctypes.data))
        return out

    def double_scalar_mult_basepoint_batch_ex(self, impl, u1, u2, points):
        """u1*G + u2*P (u1 None: u2*P) with an implementation selector (IMPL_COMPLETE / IMPL_FAST)."""
        u2 = _arr(u2, 32)
        n = u2.shape[0]
        points = _arr(points, 65, n)
        u1a = _arr(u1, 32, n) if u1 is not None else None
        out = self._points_out(n)
        self._check(self._lib.s2k_double_scalar_mult_basepoint_batch_ex(self._h, impl, n,
                                                                        u1a.ctypes.data if u1a is not None else None,
                                                                        u2.ctypes.data, points.ctypes.data, out.ctypes.data))
        return out

    def fp_op_batch_ex(self, op, operands, lazy=0):
        """Hot-path field / group arithmetic (s2k_fp_op_batch_ex, IMPL_FAST): `operands` is a list of up
        to five (n,32) operand arrays (None = unused); returns (out, out2, flag)."""
        ops = [(_arr(x, 32) if x is not None else None) for x in operands] + [None] * (5 - len(operands))
        n = ops[0].shape[0]
        ptrs = (C.c_void_p * 5)(*[(x.ctypes.data if x is not None else None) for x in ops])
        out, out2, flag = (np.zeros((n, 32), dtype=np.uint8), np.zeros((n, 32), dtype=np.uint8), np.zeros(n, dtype=np.uint8))
        self._check(self._lib.s2k_fp_op_batch_ex(self._h, IMPL_FAST, op, lazy, n, ptrs, out.ctypes.data, out2.ctypes.data,
                                                 flag.ctypes.data))
        return out, out2, flag

    def fn_split_glv_odd_batch(self, k):
        """sc_split_glv_odd: (|k1|, |k2|, signs) with both magnitudes odd and below 2^129."""
        k = _arr(k, 32)
        n = k.shape[0]
        k1, k2, sg = np.zeros((n, 32), np.uint8), np.zeros((n, 32), np.uint8), np.zeros(n, np.uint8)
        self._check(self._lib.s2k_fn_split_glv_batch_ex(self._h, IMPL_FAST, n, k.ctypes.data, k1.ctypes.data, k2.ctypes.data,
                                                        sg.ctypes.data))
        return k1, k2, sg

    def point_add_batch(self, a, b):
        a = _arr(a, 65)
        n = a.shape[0]
        b = _arr(b, 65, n)
        out = self._points_out(n)
        self._check(self._lib.s2k_point_add_batch(self._h, n, a.ctypes.data, b.ctypes.data, out.ctypes.data))
        return out

    def point_double_batch(self, a):
        a = _arr(a, 65)
        out = self._points_out(a.shape[0])
        self._check(self._lib.s2k_point_double_batch(self._h, a.shape[0], a.ctypes.data, out.ctypes.data))
        return out

    def multi_scalar_mult(self, scalars, points) -> bytes:
        """sum_i scalars[i] * points[i] as a 65-byte record (MultiScalarMultVartime)."""
        scalars = _arr(scalars, 32)
        n = scalars.shape[0]
        points = _arr(points, 65, n)            # length mismatch raises, like the reference panics
        out = np.zeros(65, dtype=np.uint8)
        self._check(self._lib.s2k_multi_scalar_mult(self._h, n, scalars.ctypes.data if n else None,
                                                    points.ctypes.data if n else None, out.ctypes.data))
        return out.tobytes()

    def multi_scalar_mult_device(self, n, d_scalars, d_points, d_out65, stream=0):
        self._check(self._lib.s2k_multi_scalar_mult_device(self._h, int(n), d_scalars, d_points, d_out65, stream))

    def point_decode_batch(self, enc, enc_len):
        enc = _arr(enc, enc_len)
        n = enc.shape[0]
        out, ok = self._points_out(n), np.zeros(n, dtype=np.uint8)
        self._check(self._lib.s2k_point_decode_batch(self._h, n, enc_len, enc.ctypes.data, out.ctypes.data, ok.ctypes.data))
        return out, ok

    # ---- field / scalar -------------------------------------------------------------
    def _field(self, fn, op, a, b):
        a = _arr(a, 32)
        n = a.shape[0]
        bb = _arr(b, 32, n) if b is not None else None
        out, flag = np.zeros((n, 32), dtype=np.uint8), np.zeros(n, dtype=np.uint8)
        self._check(fn(self._h, op, n, a.ctypes.data, bb.ctypes.data if bb is not None else None, out.ctypes.data,
                       flag.ctypes.data))
        return out, flag

    def fp_op_batch(self, op, a, b=None):
        return self._field(self._lib.s2k_fp_op_batch, op, a, b)

    def fn_op_batch(self, op, a, b=None):
        return self._field(self._lib.s2k_fn_op_batch, op, a, b)

    def fn_split_glv_batch(self, k):
        k = _arr(k, 32)
        n = k.shape[0]
        k1, k2 = np.zeros((n, 32), dtype=np.uint8), np.zeros((n, 32), dtype=np.uint8)
        self._check(self._lib.s2k_fn_split_glv_batch(self._h, n, k.ctypes.data, k1.ctypes.data, k2.ctypes.data))
        return k1, k2

    def generator_window_bits(self) -> int:
        """window bits of the generator tables this context's launches use NOW (gt_info; the library's target width is
        s2k_generator_window_bits)"""
        return self.gt_info()["bits"]

    def gtable_entry(self, i, d) -> bytes:
        out = np.zeros(64, dtype=np.uint8)
        self._check(self._lib.s2k_debug_gtable_entry(self._h, i, d, out.ctypes.data))
        return out.tobytes()


class Ticket:
    """A batch in flight (s2k_ticket): wait() blocks until its verdicts are there and returns them.  The arrays of the batch
    belong to the owner (Engine / Group) until then: dropping a Ticket is safe."""

    def __init__(self, owner, ticket, out):
        self._owner, self.ticket, self._out = owner, ticket, out
        self._done = False

    def wait(self) -> np.ndarray:
        if not self._done:
            self._owner._wait(self.ticket)
            self._done = True
            self._owner._release(self.ticket)
        return self._out

    def done(self) -> bool:
        """s2k_poll: True once the verdicts are delivered (never blocks)."""
        if not self._done and hasattr(self._owner, "_poll") and self._owner._poll(self.ticket):
            self._done = True
            self._owner._release(self.ticket)
        return self._done


def device_count() -> int:
    """Devices the HIP runtime sees (s2k_device_count); 0 without a GPU."""
    return int(load_library().s2k_device_count())


class Group(_TicketOwner):
    """Several devices behind one process (s2k_group): one context and one host thread per listed device, contiguous
    index shards, verdicts written straight into the result array."""

    def __init__(self, devices, gt_bits: int = 0, wait_tables: bool = False):
        """gt_bits / wait_tables: as Engine's (s2k_group_create_ex): every member, and its child contexts, on that table width."""
        self._lib = load_library()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self._lib.s2k_group_create_ex(devs, len(devices), int(gt_bits), CTX_WAIT_TABLES if wait_tables else 0, C.byref(h))
        if rc != 0:
            raise EngineError(f"s2k_group_create_ex failed ({rc})")
        self._h = h
        self.devices = list(devices)
        import weakref
        self._keysets = weakref.WeakSet()     # key sets go before their group (s2k_group_keyset_destroy follows the group pointer)

    def close(self):
        if getattr(self, "_h", None):
            for ks in list(self._keysets):
                ks.close()
            self._lib.s2k_group_destroy(self._h)    # (finishes the batches in flight first; frees the group's host blocks)
            self._h = None
            self._inflight = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self):
        return int(self._lib.s2k_group_size(self._h))

    def _check(self, rc):
        if rc != 0:
            raise EngineError(f"group call failed ({rc}): {self._lib.s2k_group_last_error(self._h).decode()}")

    def _wait(self, ticket):
        self._check(self._lib.s2k_group_wait(self._h, ticket))

    def set_key_grouping(self, mode: int = KEYS_AUTO, min_group: int = 0, hash_bits: int = 0, max_tables: int = 0):
        self._check(self._lib.s2k_group_set_key_grouping(self._h, int(mode), int(min_group), int(hash_bits), int(max_tables)))

    def set_small_batch_max(self, max_n: int):
        self._check(self._lib.s2k_group_set_small_batch_max(self._h, int(max_n)))

    def set_mid_batch_max(self, max_n: int):
        self._check(self._lib.s2k_group_set_mid_batch_max(self._h, int(max_n)))

    def gt_wait(self) -> int:
        """block until every member's device has its wide generator tables (s2k_group_gt_wait); the smallest width in use"""
        return int(self._lib.s2k_group_gt_wait(self._h))

    def shard_size(self, n: int) -> int:
        """items per member of a batch of n (s2k_group_shard_size)"""
        return int(self._lib.s2k_group_shard_size(self._h, int(n)))

    def host_alloc(self, n: int, width: int) -> np.ndarray:
        """An (n, width) uint8 array in page-locked memory whose shard ranges lie on the NUMA node of the member that will read
        them (s2k_group_host_alloc).  The memory belongs to the group: it is freed by host_free() or with the group, and the
        array must not be used after that."""
        p = self._lib.s2k_group_host_alloc(self._h, int(width), int(n))
        if not p:
            raise EngineError(f"s2k_group_host_alloc failed: {self._lib.s2k_group_last_error(self._h).decode()}")
        a = np.ctypeslib.as_array((C.c_uint8 * (n * width)).from_address(p)).reshape(n, width)
        return a

    def host_free(self, a: np.ndarray):
        self._lib.s2k_group_host_free(self._h, a.ctypes.data)

    def member_stats_ex(self):
        """Per member: dict(n, lo, ms, device, numa_node, bound_cpus, h2d_ms, device_ms) of its last finished shard
        (s2k_group_member_stats_ex; the device-clock times are zero until a shard has run after the first call)."""
        m = len(self)
        st = np.zeros(8 * m, dtype=np.float64)
        self._check(self._lib.s2k_group_member_stats_ex(self._h, st.ctypes.data))
        keys = ("n", "lo", "ms", "device", "numa_node", "bound_cpus", "h2d_ms", "device_ms")
        return [dict(zip(keys, st[8 * i:8 * i + 8].tolist())) for i in range(m)]

    def ecdsa_verify_batch_submit(self, pub_xy, digest32, r, s, out=None, reject_malleable: bool = False) -> Ticket:
        arrs = [(_arr(a, w) if not (isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]) else a.reshape(-1, w))
                for a, w in ((pub_xy, 64), (digest32, 32), (r, 32), (s, 32))]
        n = arrs[2].shape[0]
        if any(a.shape[0] != n for a in arrs):
            raise ValueError("length mismatch")
        out = _out_array(out, n)
        t = C.c_uint64(0)
        self._check(self._lib.s2k_group_ecdsa_verify_batch_submit(self._h, n, arrs[0].ctypes.data, arrs[1].ctypes.data,
                                                                  arrs[2].ctypes.data, arrs[3].ctypes.data,
                                                                  REJECT_MALLEABLE if reject_malleable else 0, out.ctypes.data,
                                                                  C.byref(t)))
        return self._hold(int(t.value), out, arrs)

    def ecdsa_verify_batch(self, pub_xy, digest32, r, s, reject_malleable: bool = False) -> np.ndarray:
        return self.ecdsa_verify_batch_submit(pub_xy, digest32, r, s, reject_malleable=reject_malleable).wait()

    def ecdsa_verify_encoded_batch_submit(self, pubs, digests, sigs, encoding=ENCODING_ASN1, digest_len=0,
                                          reject_malleable=False, bip0066=False) -> Ticket:
        """s2k_group_ecdsa_verify_encoded_batch_submit for lists of byte strings (or pre-built (blob, offsets) pairs)."""
        def cat(x):
            return x if isinstance(x, tuple) else _concat(list(x))
        (pb, po), (db, do), (sb, so) = cat(pubs), cat(digests), cat(sigs)
        n = len(po) - 1
        if len(do) - 1 != n or len(so) - 1 != n:
            raise ValueError("length mismatch")
        out = np.zeros(n, dtype=np.uint8)
        flags = (REJECT_MALLEABLE if reject_malleable else 0) | (BIP0066 if bip0066 else 0)
        t = C.c_uint64(0)
        self._check(self._lib.s2k_group_ecdsa_verify_encoded_batch_submit(self._h, n, pb.ctypes.data, po.ctypes.data, db.ctypes.data,
                                                                          do.ctypes.data, sb.ctypes.data, so.ctypes.data, encoding,
                                                                          digest_len, flags, out.ctypes.data, C.byref(t)))
        return self._hold(int(t.value), out, [pb, po, db, do, sb, so])

    def schnorr_batch_verify_rlc(self, pk32, msgs, sig64, seed32: bytes | None = None) -> bool:
        """BIP-340 whole-batch check with the signatures sharded over the members (s2k_group_schnorr_batch_verify_rlc)."""
        pk32 = _arr(pk32, 32)
        n = pk32.shape[0]
        sig64 = _arr(sig64, 64, n)
        seed = np.frombuffer(seed32 if seed32 is not None else os.urandom(32), dtype=np.uint8)
        res = C.c_int(0)
        if isinstance(msgs, (list, tuple)):
            offs = np.zeros(n + 1, dtype=np.uint64)
            offs[1:] = np.cumsum([len(m) for m in msgs], dtype=np.uint64)
            blob = np.frombuffer(b"".join(msgs) or b"\0", dtype=np.uint8)
            self._check(self._lib.s2k_group_schnorr_batch_verify_rlc(self._h, n, pk32.ctypes.data, blob.ctypes.data, offs.ctypes.data, 0,
                                                                     sig64.ctypes.data, seed.ctypes.data, C.byref(res)))
        else:
            m = np.ascontiguousarray(msgs, dtype=np.uint8).reshape(n, -1) if n else np.zeros((0, 0), np.uint8)
            self._check(self._lib.s2k_group_schnorr_batch_verify_rlc(self._h, n, pk32.ctypes.data, m.ctypes.data if m.size else None, None,
                                                                     m.shape[1], sig64.ctypes.data, seed.ctypes.data, C.byref(res)))
        return bool(res.value)

    def multi_scalar_mult(self, scalars, points) -> bytes:
        """sum of scalars[i] * points[i] with the terms sharded over the members (s2k_group_multi_scalar_mult) -> 65-byte record"""
        k = _arr(scalars, 32)
        n = k.shape[0]
        p = _arr(points, 65, n)
        out = np.zeros(65, dtype=np.uint8)
        self._check(self._lib.s2k_group_multi_scalar_mult(self._h, n, k.ctypes.data, p.ctypes.data, out.ctypes.data))
        return out.tobytes()

    def keyset_create(self, pub_xy, layout: int = 0) -> "GroupKeySet":
        """The tables of a fixed list of public keys on every member's device (s2k_group_keyset_create)."""
        return GroupKeySet(self, pub_xy, layout)

    def ecdsa_verify_batch_keyset_submit(self, keyset, key_index, digest32, r, s, out=None, reject_malleable: bool = False) -> Ticket:
        ki = np.ascontiguousarray(key_index, dtype=np.uint32).reshape(-1)
        n = ki.shape[0]
        arrs = [ki] + [(_arr(a, 32) if not (isinstance(a, np.ndarray) and a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]) else a.reshape(-1, 32))
                       for a in (digest32, r, s)]
        if any(a.shape[0] != n for a in arrs):
            raise ValueError("length mismatch")
        out = _out_array(out, n)
        t = C.c_uint64(0)
        self._check(self._lib.s2k_group_ecdsa_verify_batch_keyset_submit(self._h, keyset._k, n, arrs[0].ctypes.data, arrs[1].ctypes.data,
                                                                         arrs[2].ctypes.data, arrs[3].ctypes.data,
                                                                         REJECT_MALLEABLE if reject_malleable else 0, out.ctypes.data,
                                                                         C.byref(t)))
        return self._hold(int(t.value), out, arrs + [keyset])

    def ecdsa_verify_batch_keyset(self, keyset, key_index, digest32, r, s, reject_malleable: bool = False) -> np.ndarray:
        return self.ecdsa_verify_batch_keyset_submit(keyset, key_index, digest32, r, s, reject_malleable=reject_malleable).wait()

    def member_stats(self):
        """Per member, of its last finished shard: dict(n, first, ms, device)."""
        st = (C.c_double * (4 * len(self)))()
        self._check(self._lib.s2k_group_member_stats(self._h, st))
        return [{"n": int(st[4 * i]), "first": int(st[4 * i + 1]), "ms": float(st[4 * i + 2]), "device": int(st[4 * i + 3])}
                for i in range(len(self))]


class GroupKeySet:
    """Handle of s2k_group_keyset_*: the per-key tables of a fixed key list on every device of a group."""

    def __init__(self, group: "Group", pub_xy, layout: int = 0):
        pub_xy = _arr(pub_xy, 64)
        self._grp = group
        k = C.c_void_p()
        group._check(group._lib.s2k_group_keyset_create(group._h, pub_xy.shape[0], pub_xy.ctypes.data, int(layout), C.byref(k)))
        self._k = k
        group._keysets.add(self)

    def layout(self) -> int:
        return int(self._grp._lib.s2k_group_keyset_layout(self._k))

    def __len__(self):
        return int(self._grp._lib.s2k_group_keyset_size(self._k))

    def device_bytes(self):
        """per member"""
        return int(self._grp._lib.s2k_group_keyset_device_bytes(self._k))

    def close(self):
        if self._k and self._grp._h:
            self._grp._lib.s2k_group_keyset_destroy(self._k)
        self._k = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class KeySet:
    """Handle of s2k_keyset_*: the per-key tables of a fixed key list on the engine's device."""

    def __init__(self, engine: "Engine", pub_xy, layout: int = 0):
        pub_xy = _arr(pub_xy, 64)
        self._eng = engine
        k = C.c_void_p()
        engine._check(engine._lib.s2k_keyset_create_ex(engine._h, pub_xy.shape[0], pub_xy.ctypes.data, int(layout), C.byref(k)))
        self._k = k

    def layout(self) -> int:
        """KEYSET_CHUNKS or KEYSET_JOINT (s2k_keyset_layout)."""
        return int(self._eng._lib.s2k_keyset_layout(self._k))

    def __len__(self):
        return int(self._eng._lib.s2k_keyset_size(self._k))

    def device_bytes(self):
        return int(self._eng._lib.s2k_keyset_device_bytes(self._k))

    def valid_keys(self) -> np.ndarray:
        out = np.zeros(len(self), dtype=np.uint8)
        self._eng._check(self._eng._lib.s2k_keyset_valid_keys(self._k, out.ctypes.data))
        return out

    def close(self):
        if self._k:
            self._eng._lib.s2k_keyset_destroy(self._k)
            self._k = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
