# Builds the C-ABI HIP library (gfx950 only) and the oracle-side helpers.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
SRC   := $(wildcard fedmlp_amd/csrc/*.hip)
HDR   := $(wildcard fedmlp_amd/csrc/*.h) include/fedmlp_hip.h include/fedmlp_hip_debug.h
OBJ   := $(patsubst fedmlp_amd/csrc/%.hip,build/%.o,$(SRC))
LIB   := fedmlp_amd/libfedmlp_hip.so

all: $(LIB)

# make TUNING=1: the kernels' tuning knobs (common.h fm_tune) become environment switches -- for measurements only
TUNEFLAG := $(if $(TUNING),-DFM_TUNING,)

build/%.o: fedmlp_amd/csrc/%.hip $(HDR)
	@mkdir -p build
	$(HIPCC) --offload-arch=$(ARCH) -O3 -fPIC -std=c++17 -Wall -Wno-unused-function $(TUNEFLAG) -c $< -o $@

$(LIB): $(OBJ)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJ)

clean:
	rm -rf build $(LIB)

# measurement-only artefacts (tuning / probe builds, assembly dumps): nothing of the product depends on them
clean-scratch:
	rm -rf scratch tune build/*.s __pycache__ .pytest_cache

.PHONY: all clean clean-scratch
