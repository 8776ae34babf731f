import os

import numpy as np

from dftpav_amd import scenarios as sc
from dftpav_amd.pods import LayoutSpec, SurroundSet

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["cfg1", "cfg2", "cfg3", "cfg5"]


def load(name):
    """-> (Scenario rebuilt from the stored inputs, record)"""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    lay = LayoutSpec(z["piece_nums"], z["singuls"], H=4)
    sur = None
    if int(z["has_surround"]):
        sur = SurroundSet(z["sur_off"], z["sur_dur"], z["sur_coef"], z["sur_total"], z["sur_start"])
    s = sc.Scenario(name, lay, int(z["K"]), int(z["Kd"]), int(z["B"]), np.ascontiguousarray(z["ini_states"]),
                    np.ascontiguousarray(z["fin_states"]), np.ascontiguousarray(z["inner_pts"]),
                    np.ascontiguousarray(z["init_Ts"]), np.ascontiguousarray(z["corridor"]), float(z["t_now"]), 0.0, sur)
    return s, z
