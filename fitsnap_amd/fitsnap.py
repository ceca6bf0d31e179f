"""Thin orchestrator for the FIT STAGE behind the reference's ``FitSnap`` class surface
(fitsnap3lib/fitsnap.py:43-231): same attributes (``pt, config, calculator, solver, output``)
and methods (``process_configs``, ``perform_fit``, ``write_output``).

The descriptor stage needs LAMMPS, which is outside this repository: ``process_configs``
therefore either drives an injected/real ``lammps`` module through the calculators of this
package, or — the drop-in for the fit stage on a machine without LAMMPS — ingests the
reference's own on-disk hand-off (``[EXTRAS] dump_descriptors/dump_truth/dump_weights/
dump_dataframe`` -> Descriptors.npy, Truth-Ref.npy, Weights.npy, FitSNAP.df;
calculator.py:329-348) with ``load_descriptors``."""
from __future__ import annotations

import os

import numpy as np

from .config import Config
from .parallel_tools import ParallelTools
from .parallel_tools import LabelList
from .solvers.solver_factory import solver as make_solver


def row_owner(m, size, configs=None, groups=None):
    """Rank that owns each of the ``m`` rows of a dumped descriptor matrix in a job of ``size`` ranks.

    The reference deals CONFIGURATIONS round-robin (configuration i -> proc i % size,
    fitsnap3lib/parallel_tools.py:612-651) and the rows of one configuration are contiguous in A (energy row, force rows,
    stress rows: lammps_snap.py:268-330).  With the ``Configs`` (and ``Groups``) column of FitSNAP.df a configuration is
    a maximal run of rows with the same (group, file name); run i goes to rank i % size.  Without the column, or when
    the file holds fewer runs than ranks (a dump sorted by row type), the rows are cut into ``size`` contiguous blocks
    of near-equal length -- the statistics are sums over rows, any partition gives the same fit."""
    m, size = int(m), int(size)
    if size <= 1 or m == 0:
        return np.zeros(m, dtype=np.int32)
    if configs is not None and len(configs) == m:
        c = np.asarray(configs)
        change = c[1:] != c[:-1]
        if groups is not None and len(groups) == m:
            g = np.asarray(groups)
            change = change | (g[1:] != g[:-1])
        run = np.concatenate([[0], np.cumsum(change)])
        if int(run[-1]) + 1 >= size:
            return (run % size).astype(np.int32)
    return np.minimum(np.arange(m, dtype=np.int64) * size // m, size - 1).astype(np.int32)


def label_list(key, column):
    """One column of row labels as the ``list`` the reference keeps in ``pt.fitsnap_dict`` -- a ``LabelList``: a list whose
    edits are counted, so that the solvers' caches need not re-read 10^6 entries per call."""
    col = np.asarray(column)
    if key == "Testing":
        col = col.astype(bool)
    return LabelList(col.tolist())


class FitSnap:
    def __init__(self, input=None, comm=None, arglist=None):
        self.pt = ParallelTools(comm=comm)
        self.config = Config(self.pt, input, arguments_lst=list(arglist or []))
        self.calculator = None
        self.solver = make_solver(self.config.sections["SOLVER"].solver, self.pt, self.config)
        self.output = None
        if self.config.sections["CALCULATOR"].calculator.upper() == "LAMMPSSNAP" and "BISPECTRUM" in self.config.sections:
            from .io.outputs.snap import Snap

            self.output = Snap("SNAP", self.pt, self.config)
        elif self.config.sections["CALCULATOR"].calculator.upper() == "LAMMPSPACE" and "ACE" in self.config.sections:
            from .io.outputs.pace import Pace

            self.output = Pace("PACE", self.pt, self.config)
        self.data = []
        self.fit = None

    def __del__(self):
        try:
            self.pt.free()
        except Exception:
            pass

    # -- descriptor stage -----------------------------------------------------------------
    def load_descriptors(self, directory=".", shard=True):
        """Ingest Descriptors.npy / Truth-Ref.npy / Weights.npy (+ FitSNAP.df for the row
        metadata lists Testing / Row_Type / Groups / Configs / Atom_I / Atom_Type).

        Multi-rank job (one process per GPU): every rank keeps only the rows of ITS configurations -- configuration
        ``i`` belongs to rank ``i % size``, the reference's partition of the per-configuration loop
        (fitsnap3lib/parallel_tools.py:612-651) -- read through a memory map, so no rank ever holds the whole matrix.
        ``pt.local_lists`` describes this rank's rows, ``pt.fitsnap_dict`` all rows in rank order, exactly what
        ``Calculator.collect_distributed_lists`` leaves behind after the LAMMPS stage.  ``shard=False`` loads every
        row on every rank (the single-process behaviour).  Returns (rows of this rank, columns)."""
        ex = self.config.sections["EXTRAS"]
        pt = self.pt

        def path(name):
            return name if os.path.isabs(name) else os.path.join(directory, name)

        A = np.load(path(ex.descriptor_file), mmap_mode="r")
        b = np.load(path(ex.truth_file), mmap_mode="r")
        w = np.load(path(ex.weights_file), mmap_mode="r")
        if A.ndim != 2 or b.shape != (A.shape[0],) or w.shape != (A.shape[0],):
            raise ValueError("Descriptors / Truth-Ref / Weights shapes do not agree")
        m, K = A.shape
        labels = {}
        df_path = path(ex.dataframe_file)
        if os.path.exists(df_path):
            import pandas as pd

            df = pd.read_pickle(df_path)
            if len(df.index) != m:
                raise ValueError(f"{ex.dataframe_file} has {len(df.index)} rows, the descriptor matrix {m}")
            for key in ("Groups", "Configs", "Row_Type", "Atom_I", "Testing", "Atom_Type"):
                if key in df.columns:
                    labels[key] = df[key].to_numpy()
            del df
        labels.setdefault("Testing", np.zeros(m, dtype=bool))
        size, rank = (pt.get_size(), pt.get_rank()) if (shard and pt.multi) else (1, 0)
        owner = row_owner(m, size, labels.get("Configs"), labels.get("Groups"))
        self.row_owner = owner
        mine = slice(None) if size == 1 else np.flatnonzero(owner == rank)
        m_loc = m if size == 1 else int(mine.shape[0])
        pt.create_shared_array("a", m_loc, K)
        pt.create_shared_array("b", m_loc)
        pt.create_shared_array("w", m_loc)
        if m_loc:
            # fancy indexing of a memory map reads this rank's rows only
            pt.shared_arrays["a"].array.reshape(m_loc, K)[:] = A[mine]
            pt.shared_arrays["b"].array[:] = b[mine]
            pt.shared_arrays["w"].array[:] = w[mine]
        pt.new_slice_a(m_loc)
        if size == 1:
            pt.local_lists = {}
            for key, col in labels.items():
                pt.fitsnap_dict[key] = label_list(key, col)
        else:
            order = np.argsort(owner, kind="stable")             # rank-major row order = what gather_fitsnap concatenates
            pt.local_lists = {key: label_list(key, col[mine]) for key, col in labels.items()}
            for key, col in labels.items():
                pt.fitsnap_dict[key] = label_list(key, col[order])
        return m_loc, K

    def process_configs(self, data=None, allgather=False, delete_data=False):
        """fitsnap.py:134-188 with the calculators of this package (needs a ``lammps`` module)."""
        from .calculators.calculator_factory import calculator as make_calculator

        data = self.data if data is None else data
        if self.calculator is None:
            self.calculator = make_calculator(self.config.sections["CALCULATOR"].calculator, self.pt, self.config)
        self.calculator.shared_index = 0
        self.calculator.distributed_index = 0
        self.calculator.allocate_per_config(data)
        self.calculator.create_a()
        for i, configuration in enumerate(data):
            self.calculator.process_configs(configuration, i)
        if delete_data:
            del data
        self.calculator.collect_distributed_lists(allgather=allgather)
        self.calculator.extras()

    # -- fit stage (fitsnap.py:190-231) ------------------------------------------------------
    def perform_fit(self):
        if not self.config.args.perform_fit:
            return

        @self.pt.single_timeit
        def fit():
            self.solver.perform_fit()

        @self.pt.single_timeit
        def error_analysis():
            self.solver.error_analysis()

        fit()
        self.solver.fit_gather()
        has_meta = all(k in self.pt.fitsnap_dict for k in ("Groups", "Row_Type", "Testing"))
        if has_meta:
            error_analysis()
        elif self.solver.fit is not None and self.pt._rank == 0:
            bis = self.config.sections.get("BISPECTRUM")
            if self.config.sections["CALCULATOR"].calculator.upper() == "LAMMPSSNAP" and bis is not None and bis.bzeroflag:
                self.solver._offset()
        self.fit = self.solver.fit

    def write_output(self):
        if not self.config.args.perform_fit or self.output is None:
            return

        @self.pt.single_timeit
        def write_output():
            self.output.output(self.solver.fit, self.solver.errors)

        write_output()
