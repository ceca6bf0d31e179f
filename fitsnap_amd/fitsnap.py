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
from .solvers.solver_factory import solver as make_solver


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
    def load_descriptors(self, directory="."):
        """Ingest Descriptors.npy / Truth-Ref.npy / Weights.npy (+ FitSNAP.df for the row
        metadata lists Testing / Row_Type / Groups / Configs / Atom_I / Atom_Type)."""
        ex = self.config.sections["EXTRAS"]
        pt = self.pt

        def path(name):
            return name if os.path.isabs(name) else os.path.join(directory, name)

        A = np.load(path(ex.descriptor_file))
        b = np.load(path(ex.truth_file))
        w = np.load(path(ex.weights_file))
        if A.ndim != 2 or b.shape != (A.shape[0],) or w.shape != (A.shape[0],):
            raise ValueError("Descriptors / Truth-Ref / Weights shapes do not agree")
        m, K = A.shape
        pt.create_shared_array("a", m, K)
        pt.create_shared_array("b", m)
        pt.create_shared_array("w", m)
        pt.shared_arrays["a"].array[:] = A
        pt.shared_arrays["b"].array[:] = b
        pt.shared_arrays["w"].array[:] = w
        df_path = path(ex.dataframe_file)
        if os.path.exists(df_path):
            import pandas as pd

            df = pd.read_pickle(df_path)
            for key in ("Groups", "Configs", "Row_Type", "Atom_I", "Testing", "Atom_Type"):
                if key in df.columns:
                    pt.fitsnap_dict[key] = df[key].tolist()
        pt.fitsnap_dict.setdefault("Testing", [False] * m)
        return m, K

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
