"""Calculator base class: allocation of A / b / w and the row-metadata lists
(fitsnap3lib/calculators/calculator.py:13-348), with the rows resident in HBM.

Differences from the reference, by design:
  * ``create_a`` allocates the rows ON THE GPU (``fsnap_rows_alloc``) in addition to the
    host view ``pt.shared_arrays['a'|'b'|'w'].array``; the LAMMPS output of every
    configuration is transformed into its rows by the device kernel ``fsnap_assemble``
    (batched), and the host view is filled by ONE download when the lists are collected
    (``collect_distributed_lists`` — the point where ``FitSnap.process_configs`` has finished
    the per-configuration loop, fitsnap.py:134-188).  Solvers then find the rows already
    resident and skip the upload.
  * one process per GPU: every rank allocates the rows of ITS configurations only
    (reference: one node-shared array, per-proc offsets from ``new_slice_a``).
"""
from __future__ import annotations

import itertools

import numpy as np

from ..parallel_tools import DistributedList, LabelList


class Calculator:
    """Class for allocating, calculating, and collating descriptors."""

    BATCH_BYTES = 32 << 20     # raw LAMMPS bytes staged per fsnap_assemble call

    def __init__(self, name, pt, config):
        self.pt = pt
        self.config = config
        self.name = name
        self.number_of_atoms = None
        self.number_of_files_per_node = None
        self.shared_index = None
        self.distributed_index = 0
        self._batch = None
        self._rows_on_device = False

    def get_width(self):
        pass

    def create_dicts(self, nconfigs: int) -> None:
        """calculator.py:27-38."""
        self.pt.add_2_fitsnap("Groups", DistributedList(nconfigs))
        self.pt.add_2_fitsnap("Configs", DistributedList(nconfigs))
        self.pt.add_2_fitsnap("Testing", DistributedList(nconfigs))

    def allocate_per_config(self, data: list):
        """Number of atoms per configuration of THIS rank (calculator.py:42-69)."""
        n = len(data)
        self.pt.create_shared_array("number_of_atoms", n, dtype="i")
        self.pt.shared_arrays["number_of_atoms"].sliced_array = self.pt.shared_arrays["number_of_atoms"].array
        for i, configuration in enumerate(data):
            self.pt.shared_arrays["number_of_atoms"].array[i] = np.shape(configuration["Positions"])[0]

    def row_count(self):
        """a_len of the linear branch of create_a (calculator.py:263-272)."""
        calc = self.config.sections["CALCULATOR"]
        a_len = 0
        if calc.energy:
            a_len += self.number_of_atoms if calc.per_atom_energy else self.number_of_files_per_node
        if calc.force:
            a_len += 3 * self.number_of_atoms
        if calc.stress:
            a_len += self.number_of_files_per_node * 6
        return int(a_len)

    def create_a(self):
        """Allocate A, b, w (host view + HBM rows) and the row-metadata lists
        (linear branch of calculator.py:261-299)."""
        pt = self.pt
        self.number_of_atoms = int(pt.shared_arrays["number_of_atoms"].array.sum())
        self.number_of_files_per_node = len(pt.shared_arrays["number_of_atoms"].array)
        a_len = self.row_count()
        a_width = self.get_width()
        assert isinstance(a_width, int)
        a_size = a_len * a_width * pt.double_size
        ram = pt.get_ram()
        if ram and a_size / ram > 0.5 and not self.config.sections["MEMORY"].override:
            raise MemoryError("The descriptor matrix is larger than 50% of your RAM. \n Aborting...!")
        elif ram and a_size / ram > 0.5:
            pt.single_print("Warning: > 50 % RAM. I hope you know what you are doing!")
        tm = self.config.sections["SOLVER"].true_multinode
        pt.create_shared_array("a", a_len, a_width, tm=tm)
        pt.create_shared_array("b", a_len, tm=tm)
        pt.create_shared_array("w", a_len, tm=tm)
        pt.new_slice_a(a_len)
        self.shared_index = pt.fitsnap_dict["sub_a_indices"][0]
        n = pt.fitsnap_dict["sub_a_size"]
        for key in ("Groups", "Configs", "Row_Type", "Atom_I", "Testing", "Atom_Type"):
            pt.add_2_fitsnap(key, DistributedList(n))
        # rows in HBM (zero-filled) — filled by the assembly kernel as configurations arrive
        if a_len > 0:
            pt.hip().rows_alloc(a_len, a_width)
            self._rows_on_device = True
        self._batch = None

    def process_configs(self, data: dict, i: int):
        pass

    def preprocess_configs(self, data: dict, i: int):
        pass

    def preprocess_allocate(self, nconfigs: int):
        pass

    def flush_rows(self):
        """Run the assembly kernel on whatever is still staged, then fill the host view."""
        self._flush_batch()
        pt = self.pt
        if self._rows_on_device and "a" in pt.shared_arrays and pt.shared_arrays["a"].array is not None:
            sa, sb, sw = pt.shared_arrays["a"], pt.shared_arrays["b"], pt.shared_arrays["w"]
            pt.hip().download_rows(out_a=sa.array.reshape(sa.array.shape[0], -1), out_b=sb.array, out_w=sw.array)
            for arr in (sa, sb, sw):
                arr.mark_device_current()

    def _flush_batch(self):
        pass

    def collect_distributed_lists(self, allgather: bool = False):
        """calculator.py:311-326; also the end of the per-configuration loop: flush the staged
        LAMMPS blocks through the assembly kernel and download the host view."""
        self.flush_rows()
        if self.pt.stubs != 1:
            # the arrays stay rank-local (one process per GPU): keep the lists that describe them
            self.pt.local_lists = {k: LabelList(v.get_list()) for k, v in self.pt.fitsnap_dict.items()
                                   if isinstance(v, DistributedList)}
        single_process = self.pt.stubs == 1
        for key, held in list(self.pt.fitsnap_dict.items()):
            if not isinstance(held, DistributedList):
                continue
            self.pt.gather_fitsnap(key)
            gathered = self.pt.fitsnap_dict[key]
            if gathered is None:                       # a rank that does not receive the gathered lists
                continue
            # LabelList: a list (what the reference's consumers expect) that counts its edits, so the solvers need not
            # re-read every entry per call to know that a mask / category ids derived from it still hold
            if single_process:
                self.pt.fitsnap_dict[key] = LabelList(gathered.get_list())
            else:                                      # one list per rank, in rank order: concatenate
                self.pt.fitsnap_dict[key] = LabelList(itertools.chain.from_iterable(gathered))

    def extras(self):
        """Descriptors.npy / Truth-Ref.npy / Weights.npy / FitSNAP.df dumps — the on-disk
        A/b/w hand-off format (calculator.py:329-348)."""
        if self.pt._rank != 0 or "EXTRAS" not in self.config.sections:
            return
        ex = self.config.sections["EXTRAS"]
        sa = self.pt.shared_arrays
        if ex.dump_a:
            np.save(ex.descriptor_file, sa["a"].array)
        if ex.dump_b:
            np.save(ex.truth_file, sa["b"].array)
        if ex.dump_w:
            np.save(ex.weights_file, sa["w"].array)
        if ex.dump_dataframe:
            import pandas as pd

            df = pd.DataFrame(sa["a"].array)
            df["truths"] = sa["b"].array.tolist()
            df["weights"] = sa["w"].array.tolist()
            nrows = len(df.index)
            for key, labels in self.pt.fitsnap_dict.items():
                if isinstance(labels, list) and len(labels) == nrows:      # per-row label lists become columns
                    df[key] = labels
            df.to_pickle(ex.dataframe_file)
            del df
