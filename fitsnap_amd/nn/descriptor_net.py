"""Per-atom descriptor network + energy / force training step on stock PyTorch-ROCm.

BASELINE configs[4] (`examples/Ta_PyTorch_NN`) is the reference's nonlinear fit: a small softplus MLP maps the
descriptors of every atom to an atomic energy, configuration energies are their sums, forces follow from the chain
rule through the descriptor derivatives dD_i/dR_j that LAMMPS provides (`dgradflag = 1`)
(fitsnap3lib/lib/neural_networks/pytorch.py:10-48 network, :92-210 forward; solvers/pytorch.py:239-628 training loop:
Adam, weighted MSE of energies per atom and forces, `[PYTORCH]` keys layer_sizes / learning_rate / num_epochs /
batch_size / energy_weight / force_weight).  This is OUTSIDE the linear-fit hot path this repository rebuilds
("NN solver stays stock PyTorch-ROCm"), so there is nothing hand-written here: the same model and loss on
`torch.device("cuda")` (= HIP on ROCm), which is all the reference itself does (`solvers/pytorch.py:121`).  Real
descriptors need LAMMPS; `synthetic_dataset` produces tensors of the Ta example's shape (363 configurations, 30
bispectrum components per atom) from a hidden teacher network so that the fit has something to learn.
"""
from __future__ import annotations

import torch


def descriptor_network(layer_sizes):
    """MLP of the reference's shape: a first linear layer of the descriptor width (it carries the input standardisation),
    then Linear + Softplus per hidden layer and a linear output."""
    sizes = [int(n) for n in layer_sizes]
    layers = [torch.nn.Linear(sizes[0], sizes[0])]
    for a, b in zip(sizes[:-2], sizes[1:-1]):
        layers += [torch.nn.Linear(a, b), torch.nn.Softplus()]
    layers.append(torch.nn.Linear(sizes[-2], sizes[-1]))
    return torch.nn.Sequential(*layers)


class DescriptorNet(torch.nn.Module):
    """Atomic energies from descriptors; configuration energies and forces of a batch."""

    def __init__(self, layer_sizes, forces=True):
        super().__init__()
        self.net = descriptor_network(layer_sizes)
        self.forces = forces

    def forward(self, desc, config_of_atom, nconfig, dgrad=None, neighbour_of_row=None, force_slot_of_row=None, nslots=0):
        """desc (natoms, nd); config_of_atom (natoms,) long; dgrad (nrows, nd) = dD_i/dR_{j,axis} rows with
        neighbour_of_row = i and force_slot_of_row = 3 j + axis.  Returns (energies (nconfig,), forces (nslots,) or None)."""
        if self.forces:
            desc = desc.requires_grad_(True)
        e_atom = self.net(desc).squeeze(-1)
        energies = torch.zeros(nconfig, dtype=e_atom.dtype, device=e_atom.device).index_add_(0, config_of_atom, e_atom)
        if not self.forces:
            return energies, None
        de_dd = torch.autograd.grad(e_atom.sum(), desc, create_graph=True)[0]
        contrib = (dgrad * de_dd[neighbour_of_row]).sum(dim=1)                 # dE_i/dD_i . dD_i/dR_j per row
        forces = torch.zeros(nslots, dtype=e_atom.dtype, device=e_atom.device).index_add_(0, force_slot_of_row, -contrib)
        return energies, forces


def synthetic_dataset(nconfig=363, nd=30, seed=0, neighbours=20, dtype=torch.float32):
    """Tensors of the Ta example's shape; targets come from a hidden teacher network (so the loss can go down)."""
    g = torch.Generator().manual_seed(seed)
    natoms = torch.randint(2, 55, (nconfig,), generator=g)
    first = torch.cumsum(natoms, 0) - natoms
    total = int(natoms.sum())
    config_of_atom = torch.repeat_interleave(torch.arange(nconfig), natoms)
    desc = torch.randn(total, nd, generator=g, dtype=dtype) * torch.logspace(0, -2, nd, dtype=dtype)
    # every atom j feels `neighbours` atoms i of its own configuration, three Cartesian rows each
    j = torch.repeat_interleave(torch.arange(total), neighbours)
    i = first[config_of_atom[j]] + (torch.rand(j.shape[0], generator=g) * natoms[config_of_atom[j]]).long()
    rows_j = torch.repeat_interleave(j, 3)
    rows_i = torch.repeat_interleave(i, 3)
    axis = torch.arange(3).repeat(j.shape[0])
    dgrad = torch.randn(rows_j.shape[0], nd, generator=g, dtype=dtype) * 0.1
    data = dict(desc=desc, natoms=natoms, config_of_atom=config_of_atom, dgrad=dgrad, neighbour_of_row=rows_i,
                force_slot_of_row=3 * rows_j + axis, first_atom=first)
    torch.manual_seed(seed + 1)
    teacher = DescriptorNet([nd, 16, 1]).to(dtype)
    e, f = teacher(desc.clone(), config_of_atom, nconfig, dgrad, rows_i, data["force_slot_of_row"], 3 * total)
    data["energy"], data["force"] = e.detach(), f.detach()
    return data


def fit(data, layer_sizes=(30, 64, 64, 1), device="cuda", num_epochs=100, batch_size=4, learning_rate=5e-5,
        energy_weight=1e-2, force_weight=1.0, seed=0, log=None):
    """The reference's training loop shape: Adam, mini-batches of `batch_size` configurations, loss = energy_weight *
    MSE(E / natoms) + force_weight * MSE(F).  Returns (model, loss per epoch, seconds per epoch)."""
    import time

    dev = torch.device(device)
    torch.manual_seed(seed)
    model = DescriptorNet(layer_sizes).to(data["desc"].dtype).to(dev)
    try:        # one fused update kernel instead of a dozen small ones per parameter tensor (device tensors only)
        opt = torch.optim.Adam(model.parameters(), lr=learning_rate, fused=(dev.type == "cuda"))
    except (RuntimeError, TypeError):
        opt = torch.optim.Adam(model.parameters(), lr=learning_rate)
    d = {k: v.to(dev) for k, v in data.items()}
    nconfig = d["natoms"].shape[0]
    # rows of dgrad sorted by configuration: a batch of whole configurations is a contiguous slice
    cfg_of_row = d["config_of_atom"][d["force_slot_of_row"] // 3]
    order = torch.argsort(cfg_of_row, stable=True)
    dgrad, nb_row, slot_row = d["dgrad"][order], d["neighbour_of_row"][order], d["force_slot_of_row"][order]
    rows_per_cfg = torch.bincount(cfg_of_row, minlength=nconfig)
    row_first = (torch.cumsum(rows_per_cfg, 0) - rows_per_cfg).tolist() + [int(rows_per_cfg.sum())]
    atom_first = d["first_atom"].tolist() + [int(d["natoms"].sum())]
    losses, t_epochs = [], []
    for epoch in range(num_epochs):
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        acc = torch.zeros((), dtype=d["desc"].dtype, device=dev)     # summed on the device: no host sync per mini-batch
        for c0 in range(0, nconfig, batch_size):
            c1 = min(c0 + batch_size, nconfig)
            a0, a1, r0, r1 = atom_first[c0], atom_first[c1], row_first[c0], row_first[c1]
            e, f = model(d["desc"][a0:a1].clone(), d["config_of_atom"][a0:a1] - c0, c1 - c0, dgrad[r0:r1], nb_row[r0:r1] - a0,
                         slot_row[r0:r1] - 3 * a0, 3 * (a1 - a0))
            n = d["natoms"][c0:c1].to(e.dtype)
            loss = energy_weight * torch.mean(((e - d["energy"][c0:c1]) / n) ** 2) + \
                force_weight * torch.mean((f - d["force"][3 * a0:3 * a1]) ** 2)
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            acc += loss.detach() * (c1 - c0)
        epoch_loss = float(acc) / nconfig                             # the one synchronisation of the epoch
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        t_epochs.append(time.perf_counter() - t0)
        losses.append(epoch_loss)
        if log:
            log(f"epoch {epoch}: loss {losses[-1]:.6e}, {t_epochs[-1]:.3f} s")
    return model, losses, t_epochs
