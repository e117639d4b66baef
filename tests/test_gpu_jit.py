"""The hiprtc-specialised front-end (JIT "force") must give the same answers as the oracle -- the same
test bodies as the interpreter path, re-run with every sink kernel compiled from the generated source."""
import pytest

import tpch_util as T

pytestmark = pytest.mark.gpu


@pytest.fixture()
def jit(tc):
    st = tc.ctx.jit_stats()
    if not st["available"]:
        pytest.skip("hiprtc not available")
    tc.ctx.set_jit("force")
    before = tc.ctx.jit_stats()["launches"]
    yield tc
    tc.ctx.set_jit("auto")
    assert tc.ctx.jit_stats()["launches"] > before, "no JIT launch happened"


def test_q1_jit(jit):
    for n in (65, 200_000):
        li = T.gen_lineitem_device(jit, n, seed=7)
        for two_phase in (True, False):
            assert T.q1_result_to_rows(jit, T.run_q1(jit, li, two_phase=two_phase)) == T.q1_oracle_rows(n, seed=7)
    li = T.gen_lineitem_device(jit, 300_000, seed=11)
    assert T.q1_result_to_rows(jit, T.run_q1(jit, li, strategy="hash")) == T.q1_oracle_rows(300_000, seed=11)


def test_operators_jit(jit):
    import test_gpu_operators as M
    M.test_filter_exec(jit, 1000, 0.2)
    M.test_filter_exec(jit, 70_001, 0.0)
    M.test_filter_of_filter_and_projection(jit)
    M.test_projection_decimal_types_and_cast(jit)
    M.test_aggregate_single(jit, 500, 0.15)
    M.test_aggregate_single(jit, 40_000, 0.0)
    M.test_aggregate_partial_final_and_strategies(jit, "hash")
    M.test_aggregate_kat_alltypes_plain(jit)


def test_joins_sort_partition_jit(jit, tmp_path):
    import test_gpu_operators as M
    for jt in ("Inner", "Left", "Full", "RightAnti", "LeftSemi"):
        M.test_hash_join_types(jit, jt, 0.2)
    M.test_hash_join_null_equals_null_and_fused_filters(jit)
    M.test_join_then_aggregate_then_sort_pipeline(jit)
    M.test_sort_exec(jit, 10_000, 0.2)
    M.test_sort_fetch(jit)
    M.test_hash_partition(jit, 16)
    M.test_shuffle_writer_round_trip(jit, tmp_path)


def test_tpch_jit(jit):
    import test_gpu_tpch as M
    M.test_q3(jit, 120_000)
    M.test_q5(jit, 120_000)
