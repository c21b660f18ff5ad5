# Top-level build: everything is compiled in-tree for gfx950 (MI355X) only.
#   skele_raytracer_amd/lib/libskr.so   C-ABI library (include/skr.h): HIP kernels + host loader/writer
#   bin/raytracer                       the drop-in CLI (reference src/main.cpp:230-413)
#   oracle/liboracle.so, oracle/_ref/   the checkers (test infrastructure; see oracle/Makefile)
HIPCC   ?= /opt/rocm/bin/hipcc
ARCH    ?= gfx950
CSRC    := skele_raytracer_amd/csrc
LIBDIR  := skele_raytracer_amd/lib
# -ffp-contract=off: the arithmetic spec forbids implicit FMA contraction;
# float divide/sqrt stay correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
HIPFLAGS := --offload-arch=$(ARCH) -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
            -fno-fast-math -Wall -Wno-unused-function -Wno-pass-failed -Iinclude
KERNEL_SRCS := $(CSRC)/render_kernel.hip $(CSRC)/render_wave.hip $(CSRC)/render_nodes.hip $(CSRC)/render_generic.hip $(CSRC)/accumulate.hip
HOST_SRCS   := $(CSRC)/api.cpp $(CSRC)/scene_host.cpp $(CSRC)/multi_gpu.cpp
HDRS        := include/skr.h $(CSRC)/device_math.h $(CSRC)/shade_common.h $(CSRC)/render_params.h $(CSRC)/scene_host.h $(CSRC)/tri_chunks.h $(CSRC)/wave_common.h

all: lib cli oracle

lib: $(LIBDIR)/libskr.so
cli: bin/raytracer

OBJDIR := build/obj
OBJS   := $(OBJDIR)/render_kernel.o $(OBJDIR)/render_wave.o $(OBJDIR)/render_nodes.o $(OBJDIR)/render_generic.o $(OBJDIR)/accumulate.o $(OBJDIR)/api.o $(OBJDIR)/scene_host.o $(OBJDIR)/multi_gpu.o

# one object per translation unit (the kernel files take a minute or two each: `make -j4 lib`)
$(OBJDIR)/%.o: $(CSRC)/%.hip $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(EXTRA) -c -o $@ $<

$(OBJDIR)/%.o: $(CSRC)/%.cpp $(HDRS)
	@mkdir -p $(OBJDIR)
	$(HIPCC) $(HIPFLAGS) $(EXTRA) -c -o $@ -x hip $<

$(LIBDIR)/libskr.so: $(OBJS)
	@mkdir -p $(LIBDIR)
	$(HIPCC) --offload-arch=$(ARCH) -shared -fPIC -o $@ $(OBJS) -ldl -lpthread

bin/raytracer: $(CSRC)/raytracer_main.cpp include/skr.h $(LIBDIR)/libskr.so
	@mkdir -p bin
	g++ -O2 -std=c++17 -Wall -Iinclude -o $@ $(CSRC)/raytracer_main.cpp -L$(LIBDIR) -lskr -Wl,-rpath,'$$ORIGIN/../$(LIBDIR)' -Wl,-rpath,/opt/rocm/lib

oracle:
	$(MAKE) -C oracle all

asm: $(KERNEL_SRCS) $(HDRS)
	@mkdir -p build
	$(HIPCC) $(HIPFLAGS) --cuda-device-only -S -o build/render_kernel.s $(CSRC)/render_kernel.hip -Rpass-analysis=kernel-resource-usage 2> build/resource_usage.txt
	$(HIPCC) $(HIPFLAGS) --cuda-device-only -S -o build/render_wave.s $(CSRC)/render_wave.hip -Rpass-analysis=kernel-resource-usage 2>> build/resource_usage.txt
	$(HIPCC) $(HIPFLAGS) --cuda-device-only -S -o build/render_nodes.s $(CSRC)/render_nodes.hip -Rpass-analysis=kernel-resource-usage 2>> build/resource_usage.txt
	$(HIPCC) $(HIPFLAGS) --cuda-device-only -S -o build/render_generic.s $(CSRC)/render_generic.hip -Rpass-analysis=kernel-resource-usage 2>> build/resource_usage.txt

clean:
	rm -rf $(LIBDIR) bin build
	$(MAKE) -C oracle clean

.PHONY: all lib cli oracle asm clean
