"""BASELINE configs[4] (examples/Ta_PyTorch_NN) on ROCm: the reference's descriptor-network fit runs on stock
PyTorch ("C5: stock PyTorch run, report epochs/s only").  Synthetic tensors of the example's shape
(363 configurations, 30 descriptors per atom, layer sizes 30-64-64-1, Adam 5e-5, batches of 4 configurations, energy
and force loss); prints epochs per second on cuda:0.

    python examples/nn_descriptor_fit.py [--epochs 100] [--device cuda]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--device", default="cuda")
    args = ap.parse_args()
    import torch

    from fitsnap_amd.nn.descriptor_net import fit, synthetic_dataset

    if args.device == "cuda" and not torch.cuda.is_available():
        raise SystemExit("no ROCm device visible")
    data = synthetic_dataset()
    model, losses, secs = fit(data, num_epochs=args.epochs, device=args.device)
    steady = secs[2:] or secs
    print(json.dumps({"config": "Ta_PyTorch_NN shape: 363 configurations, 30 descriptors, layers 30-64-64-1, batch 4, Adam 5e-5, "
                                "energy + force loss, float32, synthetic descriptors (real ones need LAMMPS)",
                      "device": str(next(model.parameters()).device), "epochs": args.epochs,
                      "epochs_per_s": len(steady) / sum(steady), "first_loss": losses[0], "last_loss": losses[-1]}))


if __name__ == "__main__":
    main()
