"""Host logic of the perf-mode randomness (no GPU): the Philox stream ids one iteration consumes are pairwise distinct
across calls, layers, iterations and ranks, and the product's id arithmetic (step.TrainStep.stream_base, the + 8 k
+ l - 1 offsets of step.py / nets.py) equals the specification the GPU parity test restates (tests/test_gpu_step.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from test_gpu_step import perf_mode_stream_ids


def test_stream_ids_are_pairwise_distinct_over_iterations_and_ranks():
    ids = [i for it in range(4) for rank in range(8) for i in perf_mode_stream_ids(it, rank)]
    assert len(ids) == len(set(ids)) == 4 * 8 * 20


def test_product_stream_base_is_the_specified_one():
    import mocogan_chainer_amd.step as step
    for it in (0, 1, 17):
        for rank in (0, 3, 63):
            assert step.TrainStep.stream_base(it, rank) == (it * 64 + rank + 1) * 64 == perf_mode_stream_ids(it, rank)[0]
    # consecutive (iteration, rank) blocks do not overlap: a block spans STREAMS_PER_RANK ids, an iteration uses < 40
    assert max(perf_mode_stream_ids(0, 0)) - min(perf_mode_stream_ids(0, 0)) < step.TrainStep.STREAMS_PER_RANK
