"""Host I/O: the .bin loader and the result writer against the reference's formats
(include/helper.h:28-72 loadData; :441-481 save_result / save_txt)."""
import numpy as np
import pytest

from tests import cases


def test_load_bin_equals_load_data(pkg, oracle):
    raw = open(f"{cases.GOLDEN}/000000.bin", "rb").read()
    ref, n_ref = oracle.load_data(raw, 50000)                 # zero-padded to the cap
    pts, n = pkg.hostio.load_bin(f"{cases.GOLDEN}/000000.bin", 50000)
    assert n == n_ref == 34537 and pts.dtype == np.float32 and pts.shape == (n, 4)
    assert np.array_equal(pts, ref[:n])
    with pytest.raises(ValueError):
        pkg.hostio.load_bin(f"{cases.GOLDEN}/000000.bin", 1000)       # reference: "exceed" message + exit(-1)


def test_save_txt_format(pkg, tmp_path):
    rows = np.array([[1.5, -2.25, 0.125, 4.0, 1.75, 1.5, -0.5, 3.0, 0.875],
                     [10.0, 20.0, -1.0, 0.5, 0.25, 2.0, 1.570796, 9.0, 0.300001]], np.float32)
    p = tmp_path / "000000.txt"
    pkg.hostio.save_txt(str(p), rows, 12.5)
    lines = p.read_text().split("\n")
    # ofstream << fixed << setprecision(6): floats with 6 decimals, the class id as an int, ",  " separators, seconds first
    assert lines[0] == "12.500000"
    assert lines[1] == "1.500000,  -2.250000,  0.125000,  4.000000,  1.750000,  1.500000,  -0.500000,  3,  0.875000"
    assert lines[2].split(",  ")[7] == "9" and lines[2].endswith("0.300001")
    assert lines[3] == "" and len(lines) == 4
    assert pkg.hostio.format_results(np.zeros((0, 9), np.float32), 1.0) == "1.000000\n"


def test_wts_file_to_shaped_weights(pkg, tmp_path):
    """detect.load_weights: a .wts file (flat tensors, tools/gen_wts.py:86-99) -> the shaped dict DsvtPipeline consumes; extra
    state_dict keys are dropped, a tensor of the wrong size is refused."""
    w = pkg.synth.make_weights(blocks=1)
    extra = dict(w)
    extra["module.vfe.pfn_layers.0.norm.num_batches_tracked"] = np.zeros(1, np.float32)
    path = str(tmp_path / "dsvt.wts")
    pkg.synth.write_wts(path, extra)
    flat = pkg.synth.read_wts(path)
    assert all(v.ndim == 1 for v in flat.values())
    got = pkg.synth.shape_weights(flat, blocks=1)
    assert set(got) == set(w)
    for k in w:
        assert got[k].shape == w[k].shape and np.array_equal(got[k], w[k])
    bad = dict(flat); bad["module.vfe.pfn_layers.0.linear.weight"] = bad["module.vfe.pfn_layers.0.linear.weight"][:-1]
    with pytest.raises(ValueError):
        pkg.synth.shape_weights(bad, blocks=1)
    del bad["module.vfe.pfn_layers.0.linear.weight"]
    with pytest.raises(KeyError):
        pkg.synth.shape_weights(bad, blocks=1)
