"""profiles/summarize.py names rocprofv3's kernels the way bench.py does (traffic.json's keys): the form of the
precedence update sits at a different template position for each kernel family."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _summarize():
    spec = importlib.util.spec_from_file_location("tap_profiles_summarize", os.path.join(ROOT, "profiles", "summarize.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)                      # importing runs nothing: the script body is main()
    return m


def test_kernel_names_map_to_bench_names():
    s = _summarize().short_name
    # k_transition<D, G, NC, SW, MODE>: MODE & 3 = 0 fp32 copy, 1 bit shadow, 2 first step; higher bits = loop form / shape
    assert s("void k_transition<2, 8, 1, 4, 13>(long const*, float const*)") == "transition"
    assert s("void k_transition<2, 8, 1, 4, 14>(long const*, float const*)") == "transition_first"
    assert s("void k_transition<3, 32, 1, 8, 17>(long const*)") == "transition"
    assert s("void k_transition<3, 32, 1, 8, 18>(long const*)") == "transition_first"
    assert s("void k_transition<2, 8, 1, 4, 0>(long const*)") == "transition_copy"
    # round 6: 32 = in place, 64 = FULL (no code for absent inputs / idle slabs)
    assert s("void k_transition<2, 8, 1, 4, 77>(long const*, float const*)") == "transition"
    assert s("void k_transition<3, 32, 1, 8, 81>(long const*)") == "transition"
    assert s("void k_transition<2, 8, 1, 4, 105>(long const*)") == "transition"
    assert s("void k_transition_macs3<32, 1, 1, 5>(TransArgs)") == "transition"
    assert s("void k_transition_macs3<32, 1, 33, 5>(TransArgs)") == "transition"
    # k_transition_macs<G, NC, MODE[, WC]>, k_transition_macs3<G, NC, MODE[, WL]>: the width / sides come last
    assert s("void k_transition_macs<8, 1, 29, 7>(TransArgs)") == "transition"
    assert s("void k_transition_macs<8, 1, 30, 7>(TransArgs)") == "transition_first"
    assert s("void k_transition_macs<8, 1, 5>(TransArgs)") == "transition"
    assert s("void k_transition_macs3<32, 1, 5, 5>(TransArgs)") == "transition"
    assert s("void k_transition_macs3<32, 1, 6, 5>(TransArgs)") == "transition_first"
    assert s("void k_transition_macs3<32, 1, 6>(TransArgs)") == "transition_first"
    # the wave-per-container steps: <..., NC, MODE>
    assert s("void k_big_transition<false, 1, 1>(TransArgs, int)") == "transition"
    assert s("void k_macs3d_wave_transition<1, 2>(TransArgs, int, int)") == "transition_first"
    assert s('void k_rolling_step_soft<3, 32, 10>(unsigned long long*)') == "rolling_step"
    assert s("void k_rolling_window<3, 10>(unsigned long long*)") == "rolling_window"
    assert s("void k_episode<2, 8, true>(EpisodeArgs)") == "episode"
    assert s("k_dyn_bits(int, int, int, float const*, unsigned long long*, int*)") == "dyn_bits"
    assert s("void at::native::vectorized_elementwise_kernel<4>()") is None
