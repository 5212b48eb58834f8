// Verdict of the build-time disassembly check of gemm1x1_wspipe_kernel's literally named staging registers
// (isbfsar_amd/build.py::wspipe_registers_private). Compiled AFTER that check with its result; a library built any other way
// (or with llvm-objdump missing) says 0 and conv_dispatch.hip then never selects the weights-stationary kernels by itself.
#ifndef ISB_WSREG_VERIFIED
#define ISB_WSREG_VERIFIED 0
#endif
extern "C" int isb_wsreg_verified(void) { return ISB_WSREG_VERIFIED; }

// The same for mbfront8_kernel's hand-counted `s_waitcnt vmcnt(5)` (conv_mb8.hip: exactly five vector-memory operations -- four D-row
// stores and the pooled means -- may follow the next sample's LDS-DMA requests; a spill or a split store would leave input tiles in
// flight at the barrier). build.py::mbfront8_wait_counted checks the disassembly; 0 = the expand GEMM + depthwise launches run.
#ifndef ISB_MBF8_VERIFIED
#define ISB_MBF8_VERIFIED 0
#endif
namespace isb { int mbf8_verified() { return ISB_MBF8_VERIFIED; } }
