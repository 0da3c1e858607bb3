"""karpenter_amd -- MI355X (gfx950) implementation of Karpenter's provisioning / consolidation bin-packing hot
path behind a C ABI (include/karpgpu.h).  `lib` binds libkarpgpu.so, `abi` mirrors the descriptors, `workloads`
generates BASELINE.json's synthetic configs."""
