"""CPU: the slot bound of the append layout (optable_amd/engine.py append_slots) against a worst-case model of how the
kernels claim slots — per-wave chunks (k_trace_rolling) and per-workgroup chunks filled pass by pass (k_trace_pool)."""
import numpy as np
import pytest

from optable_amd.engine import append_slots


def pool_worst_case(rng, n_records, workgroups, chunk):
    """Slots a block-pool launch can claim: the records spread over the workgroups at random, every workgroup fills chunks of
    16 * chunk slots with passes of 1..64 records; a pass that does not fit the rest of a chunk leaves it as holes."""
    wg_chunk = min(16 * chunk, 1 << 19)
    share = rng.multinomial(n_records, np.ones(workgroups) / workgroups)
    claimed = 0
    for records in share:
        left_in_chunk = 0
        while records > 0:
            take = int(min(records, rng.integers(1, 65)))
            if take > left_in_chunk:  # crosses the end: the rest of the chunk is lost, a new chunk is claimed
                claimed += wg_chunk
                left_in_chunk = wg_chunk
            left_in_chunk -= take
            records -= take
    return claimed


@pytest.mark.parametrize("chunk", [64, 512, 2048])
@pytest.mark.parametrize("workgroups", [1, 7, 256])
def test_block_pool_bound_covers_the_worst_case(chunk, workgroups):
    rng = np.random.default_rng(chunk + workgroups)
    launch = {"kernel": 2, "threads": 1024, "workgroups": workgroups, "pair_queue": 2 | 4 | 16}
    for n_records in (0, 1, 63, 10_000, 1_234_567):
        bound = append_slots(n_records, launch, chunk)
        assert bound % 64 == 0
        for _ in range(20):
            assert pool_worst_case(rng, n_records, workgroups, chunk) <= bound, (n_records, workgroups, chunk)


def test_per_wave_bound_is_records_plus_one_chunk_per_wave():
    launch = {"kernel": 2, "threads": 1024, "workgroups": 256, "pair_queue": 1 | 2 | 4}
    assert append_slots(1000, launch, 512) == (1000 + 512 * 4096 + 63) // 64 * 64
    light = {"kernel": 1, "threads": 256, "workgroups": 65536, "pair_queue": 0}  # (no launch of the heavy kernels yet: the widest one)
    assert append_slots(0, light, 512) == 512 * 4096
