# Builds the gfx950 shared library (C ABI) in-tree.  `make` here, on CPU: hipcc cross-compiles.
HIPCC ?= /opt/rocm/bin/hipcc
HIPFLAGS = --offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops -Wno-unused-value
SRC = mineral_amd/csrc/ppo_capi.hip mineral_amd/csrc/replay.hip
HDR = mineral_amd/csrc/gemm.hip.h mineral_amd/csrc/chain.hip.h mineral_amd/csrc/wgrad_group.hip.h mineral_amd/csrc/split_bf16.hip.h mineral_amd/csrc/ln_wgrad0.hip.h mineral_amd/csrc/ln_fwd0.hip.h mineral_amd/csrc/ppo_tail.hip.h mineral_amd/csrc/sac_capi.hip.inc mineral_amd/csrc/shac_capi.hip.inc include/mineral_ppo.h include/mineral_replay.h include/mineral_sac.h include/mineral_shac.h
LIB = mineral_amd/libmineral_ppo.so

all: $(LIB)

$(LIB): $(SRC) $(HDR)
	$(HIPCC) $(HIPFLAGS) $(SRC) -o $@

# per-kernel register / LDS / spill table
resources: $(SRC) $(HDR)
	$(HIPCC) $(HIPFLAGS) -Rpass-analysis=kernel-resource-usage $(SRC) -o /tmp/_mrl_res.so 2>&1 | python3 tools/resource_table.py

# instrumented diagnostic build (cycle stamps inside the GEMM main loop), see tools/ktrace_run.py
ktrace: $(SRC) $(HDR)
	mkdir -p tools/bin && $(HIPCC) $(HIPFLAGS) -DMRL_KTRACE $(SRC) -o tools/bin/libmineral_ppo_ktrace.so

# the same with the main-loop ablation switches of tools/ktrace_ablate.sh compiled in
ktrace-ablate: $(SRC) $(HDR)
	mkdir -p tools/bin && $(HIPCC) $(HIPFLAGS) -DMRL_KTRACE -DMRL_KTRACE_ABLATE $(SRC) -o tools/bin/libmineral_ppo_ktrace.so

# stand-alone measurement programs (run on the GPU box: tools/bin travels with the tree)
PROBES = mfma_peak_probe mfma_overlap_probe placement_probe bf16x3_split_probe mfma_shape_probe
probes:
	mkdir -p tools/bin && for p in $(PROBES); do $(HIPCC) --offload-arch=gfx950 -O3 -std=c++17 -Xclang -target-feature -Xclang -packed-fp32-ops tools/$$p.hip -o tools/bin/$$p || exit 1; done

clean:
	rm -f $(LIB)
.PHONY: all resources ktrace ktrace-ablate probes clean
