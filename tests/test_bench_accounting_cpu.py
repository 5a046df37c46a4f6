"""bench.py's byte accounting (no GPU needed): the SURVEY 8(d) per-env-step figures the line reports as
`frac_alg_survey`, and the compulsory bytes of the implemented kernels that `roofline.achieved` prices."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("tap_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_survey_figures():
    b = _bench()
    # 108 B (2D W=5), 460 B (3D 5x5), mask/dynamic step 5 048 B (2D n=10), 15 128 B (3D n=10), 19 688 B (2D n=20)
    assert b.algorithmic_bytes(2, [5, 50], 10) == (108, 5048)
    assert b.algorithmic_bytes(3, [5, 5, 50], 10) == (460, 15128)
    assert b.algorithmic_bytes(2, [7, 100], 20)[1] == 19688


def test_compulsory_bytes_of_the_implemented_kernels():
    b = _bench()
    # the bit-shadow step writes the fp32 tensor and never reads it; of row 0 of `static` it needs
    # one float per env (round 3 counted the whole row: 3 153 / 9 585 / 11 017)
    assert b.compulsory_bytes("transition", 2, [5, 50], 10) == 3077
    assert b.compulsory_bytes("transition", 3, [5, 5, 50], 10) == 9349
    assert b.compulsory_bytes("transition", 2, [7, 100], 20) == 10861 and b.compulsory_bytes("rolling_step", 3, [5, 5, 250], 10) == 9937
    # the first step of an episode reads the fresh fp32 tensor once and has no shadow to read
    first, later = b.compulsory_bytes("transition_first", 2, [5, 50], 10), b.compulsory_bytes("transition", 2, [5, 50], 10)
    assert first - later == 30 * 20 * 4 - 20 * 8
    # the copy form moves the tensor twice and the column-sum shadow instead of the words
    assert b.compulsory_bytes("transition", 2, [5, 50], 10, bits=False) > 2 * 30 * 20 * 4
    # never more than SURVEY's figure for the bit-shadow step (it prices an fp32 read the kernel does not perform)
    for D, cs, n in ((2, [5, 50], 10), (3, [5, 5, 50], 10), (2, [7, 100], 20)):
        env, mask = b.algorithmic_bytes(D, cs, n)
        assert b.compulsory_bytes("transition", D, cs, n) < env + mask


def test_configs_name_the_baseline_workloads():
    b = _bench()
    assert set(b.CONFIGS) >= {"c2", "c3", "c4", "c5", "c6", "k6"}
    assert b.HBM_PEAK_GBS == 8000.0
