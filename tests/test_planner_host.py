"""Host logic on CPU: the round planner behind the C ABI (no GPU needed)."""
import pytest

from inplacemsdradixsort_amd import plan_first_round


def test_headline_config_plan():
    # BASELINE.json configs[1]: 2^30 u32 keys -> two 8-bit rounds, then 16 open bits for the counting leaf
    p = plan_first_round(1 << 30, 4, 0)
    assert (p["digit_width"], p["digit_shift"], p["expected_rounds"]) == (8, 24, 2)
    assert p["block_elems"] * 4 == 256 and p["leaf_count_bits"] == 16
    assert p["stripes"] * p["stripe_elems"] >= (1 << 30) > (p["stripes"] - 1) * p["stripe_elems"]
    assert p["workspace_bytes"] < 0.12 * 4 * (1 << 30)      # auxiliary memory stays a small fraction of the data


def test_small_input_is_a_single_leaf():
    p = plan_first_round(20000, 4, 0)
    assert p["digit_width"] == 0 and p["expected_rounds"] == 0 and p["leaf_capacity"] >= 20000


@pytest.mark.parametrize("n", [30000, 1 << 18, 1 << 22, 1 << 24, 1 << 27, 1 << 28, 3 << 28, 1 << 32])
def test_u32_plans_leave_at_most_16_open_bits_or_fit_lds(n):
    p = plan_first_round(n, 4, 0)
    assert 1 <= p["digit_width"] <= 8 and p["digit_shift"] == 32 - p["digit_width"]
    assert 1 <= p["expected_rounds"] <= 4
    assert p["stripe_elems"] % p["tile_elems"] == 0 or p["stripes"] == 1


def test_pairs_and_u64_plans():
    p = plan_first_round(1 << 30, 8, 8)
    assert p["digit_width"] == 8 and p["digit_shift"] == 56 and p["block_elems"] * 8 == 256 and p["leaf_count_bits"] == 14
    assert p["expected_rounds"] == 3                          # 16 Ki-pair segments exceed the pair leaf capacity
    q = plan_first_round(1 << 30, 8, 8, end_bit=32)           # config 5b after leading-bit skipping
    assert q["digit_shift"] == 24 and q["expected_rounds"] == 3
    u = plan_first_round(1 << 28, 8, 0)
    assert u["expected_rounds"] == 2


def test_bad_arguments():
    from inplacemsdradixsort_amd import MsdError
    with pytest.raises(MsdError):
        plan_first_round(1000, 2, 0)
    with pytest.raises(MsdError):
        plan_first_round(1000, 4, 0, end_bit=40)
