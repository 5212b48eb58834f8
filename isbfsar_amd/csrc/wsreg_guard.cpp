// Verdict of the build-time disassembly check of gemm1x1_wspipe_kernel's literally named staging registers
// (isbfsar_amd/build.py::wspipe_registers_private). Compiled AFTER that check with its result; a library built any other way
// (or with llvm-objdump missing) says 0 and conv_kernels.hip then never selects the weights-stationary kernels by itself.
#ifndef ISB_WSREG_VERIFIED
#define ISB_WSREG_VERIFIED 0
#endif
extern "C" int isb_wsreg_verified(void) { return ISB_WSREG_VERIFIED; }
