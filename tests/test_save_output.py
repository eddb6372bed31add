"""HDF5 output: the layout of paos/core/saveOutput.py written through libhdf5, read back through
libhdf5 and inspected with h5dump.  Parity note: the reference writes with h5py, which is absent here, and it
ships no .h5 fixture: the LAYOUT (groups, names, shapes, datatypes as h5py encodes them) is what is checked."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from paos_amd import save_output as so

pytestmark = pytest.mark.skipif(not so.hdf5_available(), reason="libhdf5 is not installed in this image")


def _retvals():
    from oracle.run_np import run as oracle_run
    from paos_amd.chains import syn20_chain, syn20_wavelength

    field = {"us": 0.0, "ut": 0.0}
    return [oracle_run(1.0, syn20_wavelength(k), 64, 4, field, syn20_chain()) for k in (0, 100)]


def test_save_output_layout_and_round_trip(tmp_path):
    ret = _retvals()[0]
    ret[20]["wfe"] = np.ma.MaskedArray(np.arange(16.0).reshape(4, 4), mask=np.eye(4, dtype=bool))
    path = str(tmp_path / "run.h5")
    so.save_output(ret, path)
    for num, rec in ret.items():
        g = f"/S{num:02d}"
        for key in ("amplitude", "phase", "wfo"):
            assert np.array_equal(so.read_dataset(path, f"{g}/{key}"), rec[key]), (num, key)
        for key in ("dx", "dy", "wl", "fratio", "wz", "distancetofocus"):
            v = so.read_dataset(path, f"{g}/{key}")
            assert v.shape == () and float(v) == float(rec[key]), (num, key)
        assert np.array_equal(so.read_dataset(path, f"{g}/extent"), np.array(rec["extent"]))
        assert so.read_dataset(path, f"{g}/propagator") == rec["propagator"]
        for name in ("ABCDt", "ABCDs"):
            assert np.array_equal(so.read_dataset(path, f"{g}/{name}/_ABCD"), rec[name]())
            assert float(so.read_dataset(path, f"{g}/{name}/_cout")) == rec[name].cout
    assert np.array_equal(so.read_dataset(path, "/S01/aperture/positions"), ret[1]["aperture"].positions)
    assert float(so.read_dataset(path, "/S01/aperture/a")) == ret[1]["aperture"].a
    assert np.array_equal(so.read_dataset(path, "/S20/wfe"), np.arange(16.0).reshape(4, 4))  # the data of the masked map
    assert so.read_dataset(path, "/info/program_name") == "paos_amd"
    assert so.read_dataset(path, "/info/file_name") == path
    # S20 has no aperture object: the reference skips None values
    with pytest.raises(RuntimeError):
        so.read_dataset(path, "/S20/aperture/positions")
    # keys_to_keep, overwrite
    so.save_output(ret, path, keys_to_keep=["wfo", "dx", "dy"])
    with pytest.raises(RuntimeError):
        so.read_dataset(path, "/S20/amplitude")
    assert np.array_equal(so.read_dataset(path, "/S20/wfo"), ret[20]["wfo"])
    with pytest.raises(OSError):
        so.save_output(ret, path, overwrite=False)
    with pytest.raises(NameError):
        so.save_output({1: {"bad": object()}}, str(tmp_path / "bad.h5"))
    so.save_output({1: {"names": ["alpha", "a-very-long-name"], "n": 3}}, str(tmp_path / "misc.h5"))
    names = so.read_dataset(str(tmp_path / "misc.h5"), "/S01/names")
    assert names.shape == (2, 1) and names.dtype == np.dtype("S10") and names[1, 0] == b"a-very-lon"
    assert int(so.read_dataset(str(tmp_path / "misc.h5"), "/S01/n")) == 3


def test_save_datacube_structure_as_h5dump_sees_it(tmp_path):
    rets = _retvals()
    path = str(tmp_path / "cube.h5")
    so.save_datacube(rets, path, ["1.0", "1.1953125"], keys_to_keep=["amplitude", "dx", "dy", "propagator"])
    for tag, ret in zip(("1.0", "1.1953125"), rets):
        assert np.array_equal(so.read_dataset(path, f"/{tag}/S20/amplitude"), ret[20]["amplitude"])
        assert so.read_dataset(path, f"/{tag}/S01/propagator") == ret[1]["propagator"]
    h5dump = shutil.which("h5dump") or "/opt/conda/bin/h5dump"
    if not os.path.exists(h5dump):
        pytest.skip("h5dump is not installed")
    text = subprocess.run([h5dump, "-H", path], capture_output=True, text=True, check=True).stdout
    for want in ('GROUP "info"', 'GROUP "1.0"', 'GROUP "1.1953125"', 'GROUP "S01"', 'GROUP "S20"', 'DATASET "amplitude"',
                 "H5T_IEEE_F64LE", "( 64, 64 ) / ( 64, 64 )", "DATASPACE  SCALAR", "H5T_CSET_UTF8", "STRSIZE H5T_VARIABLE"):
        assert want in text, want
    assert 'DATASET "wfo"' not in text and 'GROUP "ABCDt"' not in text  # dropped by keys_to_keep
