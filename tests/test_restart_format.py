"""Restart wire format (SUBROUTINE resave / the read at src/q-gcm.F:612-640): the Python host's
reader and writer against a file written by the reference toolchain in the reference's record sequence
(tests/golden/make_golden_restart.py)."""
import os

import numpy as np
import pytest

from common import GOLDEN
from qgcm_hip import preset, restart


def test_reads_reference_written_dump():
    cfg = preset("box_tiny")
    d = restart.read_restart(os.path.join(GOLDEN, "restart_box_tiny.bin"), cfg)
    g = np.load(os.path.join(GOLDEN, "restart_box_tiny_fields.npz"))
    assert d["tyrs"] == 12.5
    for k in ("po", "pom", "sst", "sstm"):
        assert np.array_equal(d[k], g[k]), k
    for k in ("ast", "astm", "hmixa", "hmixam"):
        assert d[k].shape == (cfg.nxta, cfg.nyta)


def test_writer_is_byte_identical(tmp_path):
    cfg = preset("box_tiny")
    src = os.path.join(GOLDEN, "restart_box_tiny.bin")
    d = restart.read_restart(src, cfg)
    out = tmp_path / "restart"
    restart.write_restart(str(out), cfg, **d)
    assert out.read_bytes() == open(src, "rb").read()


def test_wrong_grid_is_refused():
    with pytest.raises(ValueError, match="does not match grid"):
        restart.read_restart(os.path.join(GOLDEN, "restart_box_tiny.bin"), preset("box_small"))
