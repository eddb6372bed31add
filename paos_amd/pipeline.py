"""The caller of the hot path: lens file in, propagated wavefronts (and an HDF5 data cube) out.

Mirror of ``paos.core.pipeline.pipeline`` (reference paos/core/pipeline.py:28-230) for the part that drives
``run``: the same ``passvalue`` dictionary with the same defaults (pipeline.py:73-82), the same preparation of the
optical chains (``light_output`` keeps only the image plane, pipeline.py:111-114; ``wfe = "file,column"`` writes one
realisation of the aberration table into the ``Z1`` surface, pipeline.py:116-129), the same output file
(``save_datacube`` with one group per wavelength, pipeline.py:152-173) and the same return value (pipeline.py:224-230).

What differs is how the wavelengths are fanned out.  The reference starts ``n_jobs`` joblib worker processes, one
``run`` per wavelength (pipeline.py:139-150).  Here the wavelengths of a lens file travel through the GPU together, a
batch per launch (``run_batch``), and only the arrays somebody asked for come back over PCIe: with the default
``store_keys = "amplitude,dx,dy,wl"`` and ``return = False`` that is one N x N float64 array per saved surface
instead of the four the reference keeps in memory.  ``n_jobs`` is accepted and has no effect.  When the caller wants
the full records (``return = True``, ``store_keys = None`` or a key only ``run`` produces) every wavelength goes
through the drop-in ``run`` and the records are the reference's, key for key.

Plots (pipeline.py:175-221) are outside this package: ``plot = True`` is reported and skipped; the reference's
``plot_pop`` works on the returned records or on the saved file.

Keys beyond the reference's, all optional: ``precision`` ("fp64" | "fp32"), ``device`` (GPU ordinal), ``batch``
(wavefronts per launch; default: what a 4096^2 complex128 batch of 32 occupies, scaled with the grid).
"""
import logging
import time

from .chains import inject_wfe, read_wfe_table
from .parse_config import parse_config
from .raytrace import raytrace
from .save_output import save_datacube

logger = logging.getLogger("paos_amd")

# record keys that ``run_batch`` can deliver without the other arrays (everything else needs ``run``)
_ARRAY_OUTPUTS = {"amplitude": "amplitude", "phase": "phase", "wfo": "wfo"}
_BATCH_KEYS = {"aperture", "wz", "distancetofocus", "fratio", "dx", "dy", "wl", "extent", "propagator", "ABCDt",
               "ABCDs"} | set(_ARRAY_OUTPUTS)


def _set_defaults(passvalue):
    passvalue.setdefault("save", True)
    passvalue.setdefault("plot", False)
    passvalue.setdefault("n_jobs", 1)
    passvalue.setdefault("store_keys", "amplitude,dx,dy,wl")
    passvalue.setdefault("return", False)


def _set_up_logging(passvalue):
    level = passvalue.get("loglevel")
    if level is not None:
        if level in ("debug", "trace", "info"):
            logger.setLevel(logging.INFO if level == "info" else logging.DEBUG)
        else:
            logger.error("loglevel shall be one of debug, trace or info")
    if "logfile" in passvalue:
        logger.info("log file name: %s", passvalue["logfile"])
        logger.addHandler(logging.FileHandler(passvalue["logfile"]))


def prepare_chains(opt_chains, passvalue):
    """The chain of every wavelength as the propagation will see it (pipeline.py:107-129): with ``light_output`` only
    the surface named IMAGE_PLANE stays saved; with ``wfe = "file,column"`` the surface named Z1 gets column
    ``column`` of the aberration table (nm) as its Zernike coefficients, in 'standard' ordering, normalised, origin
    'x', behind three zeros for piston and tilts.  Chains without a Z1 are left as they are, like the reference does."""
    draw = None
    if passvalue.get("wfe") is not None:
        wfe_file, column = passvalue["wfe"].split(",")
        logger.debug("Wfe realization file: %s; column: %s", wfe_file, column)
        # the reference reads astropy's auto-named column "col<column + 4>": J, N, M come first
        draw = read_wfe_table(wfe_file.strip())[3][:, int(float(column))]
    prepared = []
    for chain in opt_chains:
        if draw is not None and any(item["name"] == "Z1" for item in chain.values()):
            chain = inject_wfe(chain, draw)
        else:
            chain = {key: dict(item) for key, item in chain.items()}
        if passvalue.get("light_output") is True:
            for item in chain.values():
                item["save"] = item["name"] == "IMAGE_PLANE"
        prepared.append(chain)
    return prepared


def _default_batch(gridsize):
    # 32 wavefronts at 4096^2 (8 GiB of complex128 fields), more of the smaller grids, never fewer than one
    return max(1, min(256, 32 * (4096 // int(gridsize)) ** 2)) if int(gridsize) <= 4096 else 8


def _propagate(pup_diameter, wavelengths_um, parameters, field, chains, passvalue, keys):
    from . import run as prun  # loads the HIP library: fails loudly without it

    precision = passvalue.get("precision", "fp64")
    device = int(passvalue.get("device", 0))
    grid, zoom = parameters["grid_size"], parameters["zoom"]
    full = passvalue["return"] or keys is None or not set(keys) <= _BATCH_KEYS
    if full:
        return [prun.run(pup_diameter, 1.0e-6 * wl, grid, zoom, field, chain, precision=precision, device=device)
                for wl, chain in zip(wavelengths_um, chains)]
    outputs = tuple(sorted(_ARRAY_OUTPUTS[k] for k in keys if k in _ARRAY_OUTPUTS))
    batch = int(passvalue.get("batch", _default_batch(grid)))
    retval = []
    for start in range(0, len(chains), batch):
        stop = min(start + batch, len(chains))
        retval.extend(prun.run_batch(pup_diameter, [1.0e-6 * wl for wl in wavelengths_um[start:stop]], grid, zoom, field,
                                     chains[start:stop], precision=precision, device=device, outputs=outputs,
                                     power=False))
    return retval


def pipeline(passvalue):
    """Run the propagation a ``passvalue`` dictionary describes and save / return its products.

    Keys (pipeline.py:38-60): ``conf`` lens file; ``output`` HDF5 file; ``save`` (True); ``plot`` (False; not
    produced here); ``n_jobs`` (1; no effect); ``store_keys`` ("amplitude,dx,dy,wl"; None = everything);
    ``return`` (False); ``light_output``; ``wfe`` "file,column"; ``debug`` (a diagnostic ray trace is logged);
    ``loglevel``, ``logfile``.  Returns the list of per-wavelength records when ``return`` is set, else None."""
    _set_up_logging(passvalue)
    _set_defaults(passvalue)

    logger.info("Parse lens file")
    pup_diameter, parameters, wavelengths, fields, opt_chains = parse_config(passvalue["conf"])
    field = fields[0]
    if passvalue.get("debug"):
        logger.debug("Perform a diagnostic ray tracing using field f1")
        for line in raytrace(field, opt_chains[0]):
            logger.debug(line)

    logger.info("Set up the POP")
    chains = prepare_chains(opt_chains, passvalue)
    keys = passvalue["store_keys"].split(",") if passvalue["store_keys"] is not None else None

    logger.info("Run the POP")
    started = time.time()
    retval = _propagate(pup_diameter, wavelengths, parameters, field, chains, passvalue, keys)
    logger.info("POP completed in %g s", time.time() - started)

    if passvalue["save"]:
        logger.info("Save POP simulation output .h5 file to %s", passvalue["output"])
        save_datacube(retval, passvalue["output"], list(map(str, wavelengths)), keys_to_keep=keys, overwrite=True)
    if passvalue["plot"]:
        logger.warning("plot = True: plots are not produced by paos_amd; run the reference's plot_pop on the output")
    return retval if passvalue["return"] else None
