"""Per-configuration fit weights (eweight / fweight / vweight): where the row weights ``w`` of
the linear fit come from (fitsnap3lib/scrapers/scrape.py:323-353, ``Scraper._weighting``)."""
from __future__ import annotations

import numpy as np

KB_EV = 0.00008617333262145   # Boltzmann constant used by the reference's scrapers (scrape.py)


def apply_weighting(data, group_entry, natoms, boltz=0.0, smartweights=False, use_force=True, use_stress=True,
                    kb=KB_EV):
    """Fill ``data['eweight'|'fweight'|'vweight'|...]`` in place from the group-table entry.

    * ``boltz == 0``: every key of the group entry containing the word 'weight' is copied
      (scrape.py:324-328);
    * ``boltz != 0``: ``eweight = exp((group eweight - E/natoms) / (kb * boltz))`` and the other
      weights are multiplied by it (:329-336);
    * ``smartweights``: divide by the group's testing/training size, ``fweight /= 3 natoms``,
      ``vweight /= 6`` (:338-353).
    """
    if boltz == 0:
        for key in group_entry:
            if "weight" in key:
                data[key] = group_entry[key]
    else:
        data["eweight"] = np.exp((group_entry["eweight"] - data["Energy"] / float(natoms)) / (kb * float(boltz)))
        for key in group_entry:
            if "weight" in key and key != "eweight":
                data[key] = data["eweight"] * group_entry[key]
    if smartweights:
        for key in group_entry:
            if "weight" in key:
                if data["test_bool"]:
                    data[key] /= group_entry["testing_size"]
                else:
                    try:
                        data[key] /= group_entry["training_size"]
                    except ZeroDivisionError:
                        data[key] = 0
        if use_force:
            data["fweight"] /= natoms * 3
        if use_stress:
            data["vweight"] /= 6
    return data
