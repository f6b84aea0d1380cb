"""CPU-side checks: the C-ABI library loads and exports everything include/eigentraj.h declares,
the host mirror of the reference interface has the reference's names/shapes, and the product
path fails loudly without a HIP device (no CPU fallback).  No kernels are launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from . import _golden as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
needs_no_gpu = pytest.mark.skipif(torch.cuda.is_available(), reason="asserts the behaviour on a box without a GPU")


def header_functions():
    src = open(os.path.join(ROOT, "include", "eigentraj.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(et_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from eigentrajectory_amd import _lib
    lib = _lib.lib()
    declared = header_functions()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"libetamd.so does not export {name}"
    assert sorted(_lib.SYMBOLS) == declared
    assert lib.et_abi_version() == _lib.ABI_VERSION == 3
    assert lib.et_compiled_arch() == b"gfx950"
    assert lib.et_status_string(0) == b"ok" and b"workspace" in lib.et_status_string(4)


def test_options_table_and_no_environment_reads():
    """include/eigentraj.h "tuning switches": et_set_option / et_get_option are the library's only mutable configuration
    -- unknown keys and values are rejected, values round-trip -- and no kernel source reads the environment (the test
    hook of one GPU test lives in libetamd_testhooks.so, not in the product library)."""
    from eigentrajectory_amd import _lib
    assert _lib.get_option("kmeans_loop") == "a" and _lib.get_option("kmeans_packed") == "1"
    with _lib.option("kmeans_loop", "chain"):
        assert _lib.get_option("kmeans_loop") == "c"
    assert _lib.get_option("kmeans_loop") == "a"
    with _lib.option("kmeans_packed_min", 262144):
        assert _lib.get_option("kmeans_packed_min") == "262144"
    for key, value in (("no_such_key", "1"), ("kmeans_loop", "x"), ("kmeans_packed", "yes"), ("kmeans_packed", "")):
        with pytest.raises(ValueError):
            _lib.set_option(key, value)
    csrc = os.path.join(ROOT, "eigentrajectory_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, name)).read(), name
    import subprocess
    syms = subprocess.run(["nm", "-D", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "et_testhook" not in syms and "et_debug" not in syms  # (stamp readers exist in -DET_EXP_* variant builds only)


def test_state_struct_layout_matches_header():
    from eigentrajectory_amd import _lib
    assert _lib.STATE_BYTES == 96
    offs = {n: getattr(_lib.KMeansState, n).offset for n, _ in _lib.KMeansState._fields_}
    assert offs == dict(max_abs_x=0, max_abs_c=8, n_total=16, frac=24, sim_frac=32, iter=40, done=48, bad_input=56,
                        error=64, inertia=72, fast_ok=80, min_nz_x_bits=88)
    lib = _lib.lib()
    assert lib.et_kmeans_partials_len(6, 20) == 6 * 20 + 20 + 2
    assert lib.et_kmeans_workspace_bytes(ctypes.c_int64(1000), 6, 20) > 1000 * 5
    assert lib.et_kmeans_workspace_bytes(ctypes.c_int64(1000), 33, 20) == 0  # d out of range
    assert lib.et_fit_gram_workspace_bytes(ctypes.c_int64(10 ** 7), 8, 12) >= 512 * (256 + 576 + 1) * 8


def test_argument_validation_without_device_work():
    """Invalid arguments are rejected on the host side before any launch."""
    from eigentrajectory_amd import _lib
    lib = _lib.lib()
    null = ctypes.c_void_p(0)
    z = ctypes.c_int64(0)
    assert lib.et_norm_project(null, null, ctypes.c_int64(-1), 8, 12, 6, null, null, null, null, 2,
                               ctypes.c_float(0), null, null, null, null, null) == 1
    assert lib.et_norm_project(null, null, z, 8, 12, 6, null, null, null, null, 2, ctypes.c_float(0), null, null, null,
                               null, null) == 0  # N == 0 is a valid no-op
    assert lib.et_norm_project(null, null, z, 2, 12, 6, null, null, null, null, 2, ctypes.c_float(0), null, null, null,
                               null, null) == 1  # T_obs < 3: obs[-3] does not exist
    assert lib.et_anchor_reconstruct_fwd(null, z, 0, 6, 8, 12, null, null, null, null, null, null, 2,
                                         ctypes.c_float(0), null, null) == 1
    assert lib.et_eigh_topk(null, 16, 6, null, null, null) == 1
    assert lib.et_eigh_topk_batch(1, null, null, null, null, null, null) == 1
    assert lib.et_eigh_topk_batch(9, null, null, null, null, null, null) == 1  # more than ET_EIGH_MAX_BATCH
    assert lib.et_eigh_topk_batch(0, null, null, null, null, null, null) == 0  # empty batch: no-op
    assert lib.et_kmeans_predict(null, z, 6, null, 20, null, null, null) == 1


@needs_no_gpu
def test_product_path_fails_loudly_without_gpu():
    from eigentrajectory_amd import TrajNorm, ops
    from eigentrajectory_amd._lib import ETLibraryError
    with pytest.raises(ETLibraryError):
        TrajNorm().calculate_params(torch.zeros(4, 8, 2))
    with pytest.raises(ETLibraryError):
        ops.kmeans_predict(torch.zeros(6, 10), torch.zeros(6, 20))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "eigentrajectory_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text and "et_oracle" not in text.replace(
                    "oracle/et_oracle.c", ""), f


def test_wrapper_has_the_reference_interface():
    """Sub-module names, state_dict keys and shapes of EigenTrajectory/model.py:16-32."""
    from eigentrajectory_amd import BatchKMeans, EigenTrajectory, ETAnchor, ETDescriptor, TrajNorm
    from eigentrajectory_amd.utils import DotDict, default_hyper_params
    hp = default_hyper_params()
    assert isinstance(hp, DotDict) and hp.k == 6 and hp.missing is None
    base = torch.nn.Linear(3, 3)
    hooks = DotDict(model_forward_pre_hook=None, model_forward=None, model_forward_post_hook=None)
    m = EigenTrajectory(base, hooks, hp)
    sd = m.state_dict()
    expect = {"ET_m_descriptor.U_obs_trunc": (16, 6), "ET_m_descriptor.U_pred_trunc": (24, 6),
              "ET_s_descriptor.U_obs_trunc": (16, 6), "ET_s_descriptor.U_pred_trunc": (24, 6),
              "ET_m_anchor.C_anchor": (6, 20), "ET_s_anchor.C_anchor": (6, 20),
              "baseline_model.weight": (3, 3), "baseline_model.bias": (3,)}
    assert {k: tuple(v.shape) for k, v in sd.items()} == expect
    g2 = G.load("g2_fit_all_scenes.npz")
    ref_sd = {k[4:]: torch.from_numpy(g2[k]) for k in g2.files if k.startswith("eth.ET_")}
    ref_sd.update({"baseline_model.weight": base.weight.data, "baseline_model.bias": base.bias.data})
    m.load_state_dict(ref_sd)  # a reference checkpoint loads unchanged
    assert torch.equal(m.ET_s_anchor.C_anchor.data, ref_sd["ET_s_anchor.C_anchor"])
    assert m.ET_m_descriptor.traj_normalizer.sca is True and m.ET_s_descriptor.traj_normalizer.sca is False
    for cls, names in ((TrajNorm, ["calculate_params", "get_params", "set_params", "normalize", "denormalize"]),
                       (ETDescriptor, ["normalize_trajectory", "denormalize_trajectory", "to_ET_space",
                                       "to_Euclidean_space", "truncated_SVD", "parameter_initialization", "projection",
                                       "reconstruction", "forward"]),
                       (ETAnchor, ["to_ET_space", "to_Euclidean_space", "anchor_generation", "forward"]),
                       (BatchKMeans, ["calculate_error", "calculate_inertia", "euc_sim", "kmeanspp",
                                      "initialize_centroids", "get_labels", "compute_centroids", "fit", "predict",
                                      "load_state_dict"])):
        for n in names:
            assert callable(getattr(cls, n)), f"{cls.__name__}.{n}"
    km = BatchKMeans(n_clusters=20)
    assert km.centroids is None and km.max_iter == 100 and km.tol == 1e-4 and km.init_mode == "kmeans++"
    km.load_state_dict({"centroids": torch.zeros(1, 6, 20)})
    assert km.centroids.shape == (1, 6, 20)
    a = ETAnchor(hp)
    c = torch.randn(6, 5, 20)
    assert torch.equal(a(c), c)  # zero anchors: identity (anchor.py:87)


def test_augment_and_metrics_helpers():
    from eigentrajectory_amd.utils import augment_trajectory, compute_batch_ade, compute_batch_fde
    z = G.load("g9_metrics.npz")
    np.testing.assert_allclose(compute_batch_ade(torch.from_numpy(z["pred"]), torch.from_numpy(z["gt"])).numpy(),
                               z["ade"], rtol=1e-6)
    np.testing.assert_allclose(compute_batch_fde(torch.from_numpy(z["pred"]), torch.from_numpy(z["gt"])).numpy(),
                               z["fde"], rtol=1e-6)
    o, p, _ = G.dataset("eth", "val")
    ao, ap = augment_trajectory(torch.from_numpy(o), torch.from_numpy(p))
    fo, fp = G.eth_fit_input()
    assert ao.shape[0] == 2 * o.shape[0] and np.array_equal(ao[len(o):, :, 1].numpy(), -o[:, :, 1])
    assert fo.shape == (70316, 8, 2) and fp.shape == (70316, 12, 2)


def test_round2_entry_points_validate_arguments_on_the_host():
    """The entry points added for the sklearn-recipe anchors, the scene path and the sharded drivers reject bad
    arguments before touching a device or RCCL."""
    from eigentrajectory_amd import _lib
    lib = _lib.lib()
    null, i64, f32 = ctypes.c_void_p(0), ctypes.c_int64, ctypes.c_float
    lib.et_kmeans_sharded_workspace_bytes.restype = ctypes.c_size_t
    assert lib.et_kmeanspp_workspace_bytes(i64(10_000), 6, 4) > 4 * 10_000 * 5
    assert lib.et_kmeanspp_workspace_bytes(i64(10_000), 6, 9) == 0          # more candidates than 2 + ln(255)
    assert lib.et_kmeanspp_workspace_bytes(i64(0), 6, 4) == 0
    assert lib.et_kmeanspp_seed(null, i64(100), 6, 20, 4, null, null, null, null, ctypes.c_size_t(0), null) == 1
    assert lib.et_center_columns(null, i64(100), 6, f32(1e-4), null, null, null, ctypes.c_size_t(0), null) == 1
    assert lib.et_scene_project(null, i64(0), 8, 6, null, null, 2, f32(0.3), null, null, null, null, null) == 0   # N == 0
    assert lib.et_scene_project(null, i64(5), 8, 6, null, null, 2, f32(0.3), null, null, null, null, null) == 1   # no obs
    assert lib.et_scene_project(null, i64(_lib.SCENE_MAX_N + 1), 8, 6, null, null, 2, f32(0.3), null, null, null, null,
                                null) == 1
    base = lib.et_kmeans_workspace_bytes(i64(4096), 6, 20)
    assert lib.et_kmeans_sharded_workspace_bytes(i64(4096), 6, 20, 8) >= base + 8 * 32
    assert lib.et_kmeans_sharded_workspace_bytes(i64(4096), 6, 20, 0) == 0
    assert lib.et_comm_info(null, null, null) == 1
    assert lib.et_comm_destroy(null) == 0                                      # NULL communicator: nothing to destroy
    assert lib.et_comm_init_rank(null, 2, 0, null) == 1
    assert lib.et_kmeans_fit_sharded(null, i64(10), i64(5), 6, 20, 100, f32(1e-4), null, null, null, null, null, null, null,
                                     null, ctypes.c_size_t(0), null, null) == 1   # N_total < N_local
    assert b"RCCL" in lib.et_status_string(6)


def test_reference_order_shard_geometry_on_the_host():
    """The shard rule of et_kmeans_fit_reforder_sharded (include/eigentraj.h) is host arithmetic: block = 4 L^3 points
    with L the level step of ATen's cascade for N_total / 4 items per lane (SumKernel.cpp: 2^max(4, ceil_log2(n) / 4));
    every rank before the last non-empty one holds whole blocks; anything else has no workspace (= is refused)."""
    import ctypes as C
    from eigentrajectory_amd import _lib as L
    from eigentrajectory_amd import ops
    lib = L.lib()
    block = lambda n: int(lib.et_kmeans_reforder_shard_block(L.i64(n), 6, 20))
    assert block(1023) == 0 and block(1024) == 4 * 16 ** 3
    assert block(4 << 19) == 4 * 16 ** 3 and block((4 << 19) + 4) == 4 * 32 ** 3  # ceil_log2(N / 4) 19 -> 20
    assert block(4 << 23) == 4 * 32 ** 3 and block((4 << 23) + 4) == 4 * 64 ** 3
    assert block(1 << 29) == 0 and int(lib.et_kmeans_reforder_shard_block(L.i64(100000), 5, 20)) == 0

    def ws(sizes, rank, K=20):
        arr = (C.c_int64 * len(sizes))(*sizes)
        return int(lib.et_kmeans_reforder_sharded_workspace_bytes(arr, len(sizes), rank, 6, K))

    for n, world in [(100000, 8), (1000000, 8), (10000000, 8), (50000, 3), (16384, 2), (1024, 4)]:
        sizes = ops.reference_order_shard_sizes(n, world)
        assert sum(sizes) == n and len(sizes) == world
        last = max(i for i, v in enumerate(sizes) if v)
        assert all(v % block(n) == 0 for v in sizes[:last]) and all(v == 0 for v in sizes[last + 1:])
        assert all(ws(sizes, r) > 0 for r in range(world))
    assert ws([25000, 25000], 0) == 0          # a cut inside a block
    assert ws([16384, 0, 1000], 1) > 0         # (an empty rank holds zero blocks wherever it sits)
    assert ws([16384, 1000], 2) == 0 and ws([16384, 1000], 0, K=33) == 0
    assert ws([16384, 16384, 0], 2) > 0        # trailing empty ranks take part in the collectives
