// The instantiations of stft_r512_fast_kernel for windows of 8 to 12, 14 and 15 rows of 32 samples and of all 512, as a translation unit of their own:
// a code object of their own, which the runtime brings to the device only when one of them is first launched (a process that never
// uses them does not pay for them at its first call).  The source is r512_fast_kernel.hip.
#define R512F_ROWS_TU 1
#include "r512_fast_kernel.hip"
