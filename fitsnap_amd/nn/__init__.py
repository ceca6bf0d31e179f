"""Descriptor-network fit on stock PyTorch-ROCm (BASELINE configs[4]): no custom kernels, outside the
linear hot path; see descriptor_net.py."""
