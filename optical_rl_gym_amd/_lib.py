"""ctypes binding of the C ABI in include/orl.h.  Fails loudly when the HIP library is missing: there is no
CPU fallback in the product."""
import ctypes as C
import os

from . import _build

_LIBS = {}


class TopologyDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_nodes", "n_links", "k_paths", "max_hops", "n_modulations")] + [
        (n, C.c_void_p) for n in ("n_paths", "path_hops", "path_links", "path_length", "path_modulation", "edge_iter_order")]


class EnvConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32)] + [(n, C.c_int32) for n in (
        "env_type", "num_spectrum_resources", "num_spatial_resources", "episode_length", "allow_rejection", "j",
        "bit_rate_mode", "bit_rate_lo", "bit_rate_hi", "n_bit_rates", "event_capacity", "action_histograms")] + [
        ("lambda_arrival", C.c_double), ("lambda_holding", C.c_double)] + [
        (n, C.c_void_p) for n in ("cum_src", "cum_dst", "bit_rates", "cum_bit_rate", "n_slots", "lmax_snr", "lmax_xt")] + [
        ("n_service_classes", C.c_int32), ("reserved", C.c_int32), ("cum_class", C.c_void_p), ("class_reward", C.c_void_p)]


class RunStats(C.Structure):
    _fields_ = [("ms_total", C.c_double), ("ms_policy", C.c_double), ("ms_step", C.c_double), ("launches", C.c_int64),
                ("n_kernels", C.c_int32), ("reserved", C.c_int32), ("ms_kernel", C.c_double * 12),
                ("kernel_name", (C.c_char * 32) * 12)]

    def kernels(self):
        """[(name, ms per launch)] of one policy+step of the device loop (time_kernels=1)."""
        return [(bytes(self.kernel_name[i].value).decode(), self.ms_kernel[i]) for i in range(self.n_kernels)]


ABI_VERSION = 2  # ORL_ABI_VERSION of include/orl.h

EXPORTS = {
    "orl_abi_version": (C.c_int, []),
    "orl_last_error": (C.c_char_p, []),
    "orl_device_count": (C.c_int, []),
    "orl_topology_create": (C.c_int, [C.POINTER(TopologyDesc), C.c_int, C.POINTER(C.c_void_p)]),
    "orl_topology_destroy": (None, [C.c_void_p]),
    "orl_batch_create": (C.c_int, [C.POINTER(EnvConfig), C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p)]),
    "orl_batch_create_seeded": (C.c_int, [C.POINTER(EnvConfig), C.c_void_p, C.c_int64, C.c_void_p, C.POINTER(C.c_void_p)]),
    "orl_batch_destroy": (None, [C.c_void_p]),
    "orl_batch_matrix_obs_dim": (C.c_int, [C.c_void_p]),
    "orl_batch_matrix_observation": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orl_batch_state_bytes": (C.c_int64, [C.c_void_p]),
    "orl_batch_get_state": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orl_batch_set_state": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orl_batch_info_dim": (C.c_int, [C.c_void_p]),
    "orl_batch_obs_dim": (C.c_int, [C.c_void_p]),
    "orl_batch_reset": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "orl_batch_policy": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "orl_batch_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orl_batch_observation": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orl_batch_get_obs_f32": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orl_batch_get_info_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "orl_batch_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orl_batch_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int, C.POINTER(RunStats)]),
    "orl_batch_sync": (C.c_int, [C.c_void_p]),
    "orl_batch_step_async": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "orl_batch_step_wait": (C.c_int, [C.c_void_p]),
    "orl_batch_set_info_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "orl_batch_policy_step": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "orl_batch_get_counters": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orl_batch_get_services": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orl_batch_get_slots": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "orl_batch_get_link_stats": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "orl_batch_get_spectrum": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "orl_batch_get_net_stats": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "orl_batch_get_active": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orl_batch_get_flags": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orl_batch_debug_stream_read": (C.c_int64, [C.c_void_p, C.c_int]),
    "orl_debug_stream_peak": (C.c_int, [C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "orl_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "orl_host_free": (C.c_int, [C.c_void_p]),
    "orl_batch_device_buffer": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "orl_batch_debug_serial_count": (C.c_int64, [C.c_void_p]),
    "orl_batch_debug_persist_spec": (C.c_int, [C.c_void_p]),
    "orl_batch_debug_persist_form": (C.c_int, [C.c_void_p]),
    "orl_batch_debug_step_kernel": (C.c_int, [C.c_void_p]),
    "orl_build_has_alt": (C.c_int, []),
    "orl_batch_debug_prof": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "orl_batch_reseed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "orl_batch_set_paths": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orl_batch_episode_log": (C.c_int, [C.c_void_p, C.c_int32]),
    "orl_batch_get_episode_log": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "orl_batch_get_episode_rewards": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orl_batch_check": (C.c_int, [C.c_void_p]),
    "orl_batch_get_action_histograms": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p]),
    "orl_batch_get_pending": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p]),
    "orl_batch_totals": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "orl_batch_row_words": (C.c_int, [C.c_void_p]),
    "orl_batch_map_words": (C.c_int, [C.c_void_p]),
    "orl_batch_get_slots_packed": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orl_batch_get_link_stats_all": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orl_batch_get_net_stats_all": (C.c_int, [C.c_void_p, C.c_void_p]),
    "orl_batch_spec_flags": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "orl_spec_flags_for": (C.c_int, [C.POINTER(EnvConfig), C.POINTER(TopologyDesc), C.c_char_p, C.c_int]),
    "orl_spec_flags_for_batch": (C.c_int, [C.POINTER(EnvConfig), C.POINTER(TopologyDesc), C.c_int64, C.c_char_p, C.c_int]),
    "orl_batch_load_spec": (C.c_int, [C.c_void_p, C.c_char_p]),
    "orl_multi_create": (C.c_int, [C.POINTER(EnvConfig), C.POINTER(TopologyDesc), C.c_int64, C.c_void_p, C.c_int, C.c_void_p,
                                   C.POINTER(C.c_void_p)]),
    "orl_multi_n_shards": (C.c_int, [C.c_void_p]),
    "orl_multi_shard": (C.c_void_p, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "orl_multi_run": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    "orl_multi_destroy": (None, [C.c_void_p]),
}


class OrlError(RuntimeError):
    pass


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  If this library pulled in the system copies
    first, a later `import torch` would bring a second HIP runtime into the process and find "No HIP GPUs"; so when torch
    is installed (it need not be imported), its copies are loaded first and both sides share one runtime — the order
    bench.py and the tests have anyway by importing torch before this package."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        cand = os.path.join(libdir, name)
        if os.path.exists(cand):
            try:
                C.CDLL(cand, mode=C.RTLD_GLOBAL)
            except OSError:
                return


def lib(variant=None):
    """Load (building if the sources changed) liborlgpu.so.  variant "alt" (or ORL_LIB_VARIANT=alt in the environment
    when the batch is created) is the -DORL_ALT_IMPLS build the cross-implementation tests use."""
    variant = variant or os.environ.get("ORL_LIB_VARIANT", "default")
    if variant not in _LIBS:
        path = _build.lib_path(variant)
        if _build.stale(variant):
            path = _build.build(variant=variant)
        if not os.path.exists(path):
            raise OrlError("%s is missing and could not be built; the HIP path is the only path" % os.path.basename(path))
        _share_hip_runtime_with_torch()
        handle = C.CDLL(path)
        for name, (res, args) in EXPORTS.items():
            fn = getattr(handle, name)  # AttributeError = ABI mismatch, fail loudly
            fn.restype = res
            fn.argtypes = args
        if handle.orl_abi_version() != ABI_VERSION:
            raise OrlError("%s reports ABI version %d, this binding is written for %d (include/orl.h)"
                           % (os.path.basename(path), handle.orl_abi_version(), ABI_VERSION))
        _LIBS[variant] = handle
    return _LIBS[variant]


def check(rc, handle=None):
    """0 -> nothing; else the exception the reference would have raised: ORL_E_ACTION -> IndexError (rmsa_env.py:167),
    ORL_E_OVERFLOW -> OverflowError (the reference's heap is unbounded), anything else -> OrlError."""
    if rc != 0:
        msg = (handle or lib()).orl_last_error()
        text = "liborlgpu: error %d: %s" % (rc, msg.decode() if msg else "?")
        if rc == -3:
            raise IndexError(text)
        if rc == -4:
            raise OverflowError(text)
        if rc == -5 and "out of host memory" in text:
            raise MemoryError(text)
        raise OrlError(text)
