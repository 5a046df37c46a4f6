"""Live comparison of the host-side mirror (PACKDataset, file format) with the imported reference.
Build container only; skipped wherever /root/reference is absent."""
import numpy as np
import pytest
import torch

import golden_util as G
import ref_loader
import tap_net_amd as T
from tap_net_amd import datafiles

pytestmark = [pytest.mark.reference,
              pytest.mark.skipif(not ref_loader.available(), reason="reference checkout not present")]


@pytest.fixture(scope="module")
def dirs(tmp_path_factory):
    out = {}
    for D, N in ((2, 256), (3, 64)):
        z = G.load("dataset_%dd.npz" % D)
        pos = z["txt_pos"].reshape(N, D, 10).transpose(0, 2, 1)
        d = str(tmp_path_factory.mktemp("ds%d" % D)) + "/"
        datafiles.write_dataset(d, z["static"], z["dynamic"], pos, container_ids=z["txt_container"])
        # a second, different directory (samples reversed) to exercise mix_data_file
        d2 = str(tmp_path_factory.mktemp("mix%d" % D)) + "/"
        datafiles.write_dataset(d2, z["static"][::-1], z["dynamic"][::-1], pos[::-1], container_ids=z["txt_container"][::-1])
        out[D] = (d, d2, N)
    return out


@pytest.mark.parametrize("D", [2, 3])
@pytest.mark.parametrize("input_type,allow_rot", [("bot", True), ("bot", False), ("simple", False), ("rot", True),
                                                  ("bot-rot", True), ("mul", True), ("mul-with", True), ("rot-old", True)])
def test_packdataset_matches_reference(dirs, D, input_type, allow_rot):
    pack = ref_loader.load()[1]
    d, d2, N = dirs[D]
    for hm_type in ("diff", "full"):
        for mix in (None, d2):
            try:
                ref = pack.PACKDataset(d, 10, N, 7, input_type, hm_type, allow_rot, 5, mix_data_file=mix, unit=1)
            except Exception as e:      # e.g. 'bot' without rotations: the reference cannot build it either
                with pytest.raises(type(e)):
                    T.PACKDataset(d, 10, N, 7, input_type, hm_type, allow_rot, 5, mix_data_file=mix, unit=1)
                continue
            mine = T.PACKDataset(d, 10, N, 7, input_type, hm_type, allow_rot, 5, mix_data_file=mix, unit=1)
            assert torch.equal(ref.static, mine.static), (input_type, allow_rot, hm_type, mix is not None)
            assert torch.equal(ref.dynamic, mine.dynamic)
            assert ref.decoder_static.shape == mine.decoder_static.shape
            assert ref.decoder_dynamic.shape == mine.decoder_dynamic.shape
            assert len(ref) == len(mine)
            for a, b in zip(ref[5], mine[5]):
                assert torch.equal(a, b)


def test_no_precedence_and_unit(dirs):
    pack = ref_loader.load()[1]
    d, _, N = dirs[2]
    ref = pack.PACKDataset(d, 10, N, 3, "bot", "diff", True, 5, unit=0.5, no_precedence=True)
    mine = T.PACKDataset(d, 10, N, 3, "bot", "diff", True, 5, unit=0.5, no_precedence=True)
    assert torch.equal(ref.static, mine.static) and torch.equal(ref.dynamic, mine.dynamic)
    assert ref.decoder_dynamic.shape == mine.decoder_dynamic.shape


def test_dataset_directory_names_match_reference(tmp_path, monkeypatch):
    """create_dataset / create_dataset_gt / get_mix_dataset name their directories as the reference does
    (pack.py:475-505, 568-608): with the directories already present both sides return without generating."""
    import os
    from tap_net_amd import pack as tpack
    _, rpack, _ = ref_loader.load()
    monkeypatch.chdir(tmp_path)
    for D in (2, 3):
        for n, tr, va, iw, sr in ((10, 1000, 100, 7, [1, 5]), (20, 64, 8, 5, [2, 6])):
            for kind in ("rand", "gt"):
                for split, size in (("train", tr), ("valid", va)):
                    d = "./data/%s_%dd/pack-%s-%d-%d-%d-%d-%d/" % (kind, D, split, n, size, iw, sr[0], sr[1])
                    os.makedirs(d, exist_ok=True)
                    open(d + "blocks.txt", "w").write("0\n")
            assert tpack.create_dataset(n, tr, va, D, iw, 50, 1, sr, seed=3) == rpack.create_dataset(n, tr, va, D, iw, 50, 1, sr, seed=3)
            assert tpack.create_dataset_gt(n, tr, va, D, 5, 50, iw, 50, "bot", 1, sr, seed=3) == \
                rpack.create_dataset_gt(n, tr, va, D, 5, 50, iw, 50, "bot", 1, sr, seed=3)
            assert tpack.get_mix_dataset(n, tr, va, D, iw, sr, seed=3) == rpack.get_mix_dataset(n, tr, va, D, iw, sr, seed=3)


@pytest.mark.parametrize("D", [2, 3])
def test_rolling_dataset_reads_what_the_reference_reads(dirs, D):
    """rolling.RollingDataset (rolling.py:462-536): the blocks / positions our reader hands to RollingWindows are
    the arrays the reference hands to each InitialContainer, and the zero decoder inputs have its shapes."""
    import importlib, sys
    from tap_net_amd.rolling import RollingDataset
    ref_loader.load()
    sys.path.insert(0, ref_loader.REFERENCE_DIR)
    try:
        rrolling = importlib.import_module("rolling")
    finally:
        sys.path.remove(ref_loader.REFERENCE_DIR)
    d, _, N = dirs[D]
    M = 6
    ref = rrolling.RollingDataset(d, 10, 5, M, D, 3, "bot", "diff", True, 5, 7, 50)
    blocks, positions = RollingDataset.read_instances(d, 10, D, M)
    for i in range(M):
        ic = ref.initial_containers[i]
        assert np.array_equal(np.asarray(ic.blocks)[:10], blocks[i]) and np.array_equal(np.asarray(ic.positions), positions[i])
    assert tuple(ref.decoder_static.shape) == (1, D, 1)
    assert tuple(ref.decoder_dynamic.shape) == ((1, 4, 1) if D == 2 else (1, 2, 5, 5))
