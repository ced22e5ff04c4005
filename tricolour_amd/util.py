"""Host-side string helpers needed by the strategy steps."""
import re

import numpy as np


def casa_style_range(val):
    """CASA-style range "lo~hi[m]" -> [lo, hi]; "" or "*" -> (0, inf).
    Same accepted grammar and errors as ``tricolour.util.casa_style_range``
    (tricolour/util.py:78-95)."""
    if not isinstance(val, str):
        raise ValueError("Value must be a string")
    if val.strip() == "" or val.strip() == "*":
        return (0, np.inf)
    number = r"(\d+(\.\d*)?|\.\d+)([eE][+-]?\d+)?"
    if re.match(r"^" + number + r"~" + number + r"[\s]*[m]?$", val):
        val = val.replace(" ", "").replace("\t", "").replace("m", "")
        return list(map(float, val.split("~")))
    raise ValueError("Value must be range or blank")
