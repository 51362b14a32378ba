"""Host-side GEMM planning (csrc/kernels_gemm.hip plan_tiles / plan_tile / plan_fullk through aprilx_plan_gemm; no GPU):
the engine decides "fused row epilogue or partial planes + row kernel" with gemm_fullk / gemm_partials and the launch plans
the kernel from the same functions -- here the answers are checked for sanity over a grid of shapes, batch sizes, z-batch
counts and schedule families, so that a planner edit that aborts, returns an impossible plane count or forgets the forced
form fails on the CPU."""
import ctypes as C
import itertools

from april_asr_amd import _ffi


def plan(M, N, kz, zc, tile_ok, force):
    out = (C.c_int32 * 3)()
    assert _ffi.lib().aprilx_plan_gemm(M, N, kz, zc, tile_ok, force, out) == 0
    return int(out[0]), int(out[1]), int(out[2])


def test_plans_are_sane(built):
    Ms = [1, 3, 16, 17, 31, 32, 33, 48, 64, 100, 250, 256, 300, 512, 1000, 1024, 2048, 4096, 5000, 8192]
    for M, N, kz, zc, tile_ok, force in itertools.product(Ms, [64, 192, 512, 768], [1, 2, 4, 8], [1, 2, 3, 12], [0, 1, 2], [0, 1]):
        fused, planes, tiled = plan(M, N, kz, zc, tile_ok, force)
        assert planes in (1, 2, 4, 8) and planes <= kz, (M, N, kz, zc, tile_ok, force, planes)
        if force and N % 32 == 0:
            assert fused == 1, (M, N, kz, zc, tile_ok)                 # the layer-major paths rely on the forced fused form
        if kz == 1:
            assert planes == 1


def test_tile_rule_starts_where_documented(built):
    # no GM_TILE below 256 32-row tiles per launch; aprilv0 N = 512: 256 rows x 3 problems = 192 tiles -> no,
    # 1024 rows x 1 -> 256 tiles -> yes; never below 32 rows; N must be a multiple of 64
    assert plan(256, 512, 8, 3, 1, 0)[2] == 0
    assert plan(1024, 512, 8, 1, 1, 0)[2] == 1
    assert plan(16, 512, 8, 12, 1, 0)[2] == 0
    assert plan(4096, 544, 8, 1, 1, 0)[2] == 0
    # fused once the launch holds 512 tiles (2048 rows x 2 problems of N = 512: 64-row tiles)
    assert plan(2048, 512, 8, 2, 1, 0)[0] == 1


def test_always_tile_plans_every_batch_size(built):
    # the fp16 tile path: one session runs the same chains as 512 (split-K planes at small M, fused at large M)
    for M in (1, 2, 7, 16, 40, 512, 4096):
        fused, planes, _ = plan(M, 768, 2, 1, 2, 0)
        assert (fused == 1) == (planes == 1) or fused == 0
    assert plan(1, 512, 8, 1, 2, 0)[0] == 0 and plan(1, 512, 8, 1, 2, 0)[1] == 8


def stream_form(kind, M, N, K, kz, groups):
    return int(_ffi.lib().aprilx_stream_form(kind, M, N, K, kz, groups))


def pick_kz(K, N):
    """csrc/engine.cc pick_kz (K slabs of a row-epilogue GEMM)"""
    kz = max(1, min(256 // max(1, N // 16), 8))
    while kz > 1 and ((K // 16) // kz < 4 or (K // 16) % (4 * kz) != 0):
        kz >>= 1
    return kz


def test_stream_kernels_take_every_layer_gemm_up_to_16_rows(built):
    """csrc/kernels_recur.hip: at <= 16 rows all six layer-GEMM forms of the four test models run as weight streams (so that
    tests/test_gpu_recur_kernels.py compares what it says it compares), above 16 rows none does."""
    from april_asr_amd import synth_model as SM
    for dims in (SM.TINY_DIMS, SM.MEDIUM_DIMS, SM.APRILV0_DIMS, SM.LARGE_DIMS):
        d, h, f = dims["d_model"], dims["hidden"], dims["ffn"]
        G = d // 32
        shapes = [(0, 4 * h, 2 * d, 1, 3), (1, 4 * h, 2 * d, 1, 1), (2, 4 * h, 2 * d, 1, 4), (3, f, d, 1, 5),
                  (4, d, h, pick_kz(h, d), 2), (5, d, f, pick_kz(f, d), 6)]
        for kind, N, K, kz, want in shapes:
            for M in (1, 2, 7, 10, 16):
                assert stream_form(kind, M, N, K, kz, G) == want, (dims["d_model"], kind, M)
            for M in (17, 64, 256):
                assert stream_form(kind, M, N, K, kz, G) == 0, (dims["d_model"], kind, M)
    # shapes without a kernel fall back: K not a multiple of 64, more than 32 sum-of-squares partials per row
    assert stream_form(3, 1, 2048, 544, 1, 16) == 0
    assert stream_form(4, 1, 2048, 1024, 4, 64) == 0
