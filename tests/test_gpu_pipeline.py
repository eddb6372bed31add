"""``pipeline`` end to end on the GPU (the caller of ``run``, pipeline.py:28-230): a shipped lens file
in, the HDF5 cube out, its arrays against the oracle run on the same prepared chains.

Aperture-mask VALUES are parity-unpinned (photutils is absent): the figures here are "the reference's
arithmetic given the builder's masks".  The HDF5 layout is checked by reading it back (h5py is absent: no byte pin)."""
import os

import numpy as np
import pytest

from conftest import rel_err

from paos_amd import parse_config, pipeline as pl
from paos_amd import save_output as so

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not so.hdf5_available(), reason="libhdf5 is not installed in this image")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LENS = os.path.join(ROOT, "data", "lens")
WFE = os.path.join(ROOT, "data", "wfe", "wfe_realization_SN20210914.csv")
TOL = 1e-10


def _oracle_records(conf, passvalue):
    from oracle.run_np import run as oracle_run

    pup, par, wls, fields, chains = parse_config(conf)
    chains = pl.prepare_chains(chains, passvalue)
    return wls, [oracle_run(pup, 1.0e-6 * wl, par["grid_size"], par["zoom"], fields[0], chain) for wl, chain in zip(wls, chains)]


def test_airs_ch0_default_cube_vs_oracle(tmp_path):
    """Ariel_AIRS-CH0.ini as shipped (512^2, four wavelengths) with the default store_keys, two wavelengths per launch:
    /<wavelength>/S##/{amplitude, dx, dy, wl} of every saved surface."""
    conf = os.path.join(LENS, "Ariel_AIRS-CH0.ini")
    out = str(tmp_path / "airs.h5")
    pv = {"conf": conf, "output": out, "batch": 2, "n_jobs": 4}
    assert pl.pipeline(pv) is None
    wls, ref = _oracle_records(conf, pv)
    assert len(wls) == 4
    for wl, rec in zip(wls, ref):
        assert len(rec) > 3
        for num, r in rec.items():
            g = f"/{wl}/S{num:02d}"
            assert rel_err(so.read_dataset(out, g + "/amplitude"), r["amplitude"]) < TOL, (wl, num)
            for key in ("dx", "dy", "wl"):
                assert float(so.read_dataset(out, f"{g}/{key}")) == r[key], (wl, num, key)
            with pytest.raises(RuntimeError):
                so.read_dataset(out, g + "/wfo")


def test_ta_ground_with_a_wfe_realisation_returned_and_saved(tmp_path):
    """lens_file_TA_Ground.ini (its Z1 surface is live) with column 5 of the aberration table injected, light output,
    everything stored and returned: the records are ``run``'s, the file holds them, both against the oracle."""
    conf = os.path.join(LENS, "lens_file_TA_Ground.ini")
    out = str(tmp_path / "ta.h5")
    pv = {"conf": conf, "output": out, "wfe": f"{WFE},5", "light_output": True, "store_keys": None, "return": True,
          "debug": True, "loglevel": "debug"}
    ret = pl.pipeline(pv)
    wls, ref = _oracle_records(conf, pv)
    assert isinstance(ret, list) and len(ret) == len(wls) == len(ref)
    for wl, got, rec in zip(wls, ret, ref):
        assert sorted(got) == sorted(rec) and len(rec) == 1  # the image plane only
        (num,) = rec
        g = f"/{wl}/S{num:02d}"
        assert rel_err(got[num]["wfo"], rec[num]["wfo"]) < TOL
        assert rel_err(got[num]["amplitude"] ** 2, rec[num]["amplitude"] ** 2) < TOL
        assert np.array_equal(so.read_dataset(out, g + "/wfo"), got[num]["wfo"])
        assert np.array_equal(so.read_dataset(out, g + "/amplitude"), got[num]["amplitude"])
        assert np.array_equal(so.read_dataset(out, g + "/ABCDt/_ABCD"), rec[num]["ABCDt"]())
        for key in ("dx", "dy", "wl", "fratio", "wz", "distancetofocus"):
            assert float(so.read_dataset(out, f"{g}/{key}")) == rec[num][key] == got[num][key], (wl, key)
        assert so.read_dataset(out, g + "/propagator") == rec[num]["propagator"]
    # the realisation matters: without it the image differs
    plain = pl.pipeline({"conf": conf, "save": False, "light_output": True, "return": True})
    (num,) = plain[0]
    assert rel_err(plain[0][num]["amplitude"], ret[0][num]["amplitude"]) > 1e-3
