"""Glass refractive indices at parse time (host scalars).

Restates what ``parse_config`` needs from paos/util/material.py (Sellmeier-1
:36-62, thermal model :64-91, air index :93-134, ``nmat`` :136-169) and the
seven-glass catalogue of paos/util/lib.py:5-97.  The catalogue numbers are
physical constants (Handbook of Optics, SCHOTT / ADEPT-INFRARED AGF files), laid
out here as one row per glass: (K1, L1, K2, L2, K3, L3, D0, Tref).
Plot helpers of the reference class are out of scope.
"""
import numpy as np

_GLASS = {
    #            K1               L1               K2               L2               K3              L3              D0        Tref
    "CAF2":     (5.67588800e-01, 2.52643000e-03, 4.71091400e-01, 1.00783330e-02, 3.84847230e+00, 1.20055600e+03, -2.6600e-05, 20.0),
    "SAPPHIRE": (1.023798000e+00, 3.775880000e-03, 1.058264000e+00, 1.225440000e-02, 5.280792000e+00, 3.213616000e+02, 1.80000000e-05, 20.0),
    "ZNSE":     (4.29801490e+00, 3.68881960e-02, 6.27765570e-01, 1.43476258e-01, 2.89556330e+00, 2.20849196e+03, 5.5400e-05, 20.0),
    "BK7":      (1.03961212e+00, 6.00069867e-03, 2.31792344e-01, 2.00179144e-02, 1.01046945e+00, 1.03560653e+02, 1.8600e-06, 20.0),
    "SF6":      (1.724484820e+00, 1.348719470e-02, 3.901048890e-01, 5.693180950e-02, 1.045728580e+00, 1.185571850e+02, 6.69000000e-06, 20.0),
    "SF11":     (1.73848403e+00, 1.36068604e-02, 3.11168974e-01, 6.15960463e-02, 1.17490871e+00, 1.21922711e+02, 1.1200e-05, 20.0),
    "BAF2":     (6.43356000e-01, 3.34000000e-03, 5.06762000e-01, 1.20300000e-02, 3.82610000e+00, 2.15169810e+03, -4.4600e-05, 20.0),
}

materials = {
    name: {
        "Tref": row[7],
        "sellmeier": dict(zip(("K1", "L1", "K2", "L2", "K3", "L3"), row[:6])),
        "Tmodel": {"D0": row[6]},
    }
    for name, row in _GLASS.items()
}


class Material:
    """``Material(wl_micron, Tambient, Pambient).nmat(name) -> (n@Tref, n@Tambient)``."""

    def __init__(self, wl, Tambient=-218.0, Pambient=1.0, materials=None):
        self.wl = wl
        self.Tambient = Tambient
        self.Pambient = Pambient
        self.materials = globals()["materials"] if materials is None else materials

    def sellmeier(self, par):
        """n^2 - 1 = sum_i K_i l^2 / (l^2 - L_i), accumulated term by term
        (material.py:57-62)."""
        wl2 = self.wl**2
        acc = par["K1"] * wl2 / (wl2 - par["L1"])
        acc += par["K2"] * wl2 / (wl2 - par["L2"])
        acc += par["K3"] * wl2 / (wl2 - par["L3"])
        return np.sqrt(acc + 1.0)

    @staticmethod
    def nT(n, D0, delta_T):
        """n + (n^2-1)/(2n) D0 dT (material.py:89-91)."""
        return n + (n**2 - 1.0) / (2.0 * n) * D0 * delta_T

    def nair(self, T, P=1.0):
        """Kohlrausch air index scaled to temperature/pressure (material.py:125-134)."""
        wl2 = self.wl**2
        nref = 1.0 + 1.0e-8 * (
            6432.8 + 2949810.0 * wl2 / (146.0 * wl2 - 1.0) + 25540.0 * wl2 / (41.0 * wl2 - 1.0)
        )
        return 1.0 + (nref - 1.0) * P / (1.0 + 3.4785e-3 * (T - 15))

    def nmat(self, name):
        glass = self.materials[name.upper()]
        n_ref = self.sellmeier(glass["sellmeier"]) * self.nair(T=glass["Tref"], P=self.Pambient)
        n_amb = self.nT(n_ref, glass["Tmodel"]["D0"], self.Tambient - glass["Tref"])
        return n_ref, n_amb
