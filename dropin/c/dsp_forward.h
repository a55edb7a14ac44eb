/* The reference has no headers: every tool of it textually includes the .c file that holds a function
 * (src/stft.cpp:16 does #include "stft.c", compiled with -Ic, Makefile:25), and those files wrap their functions in
 * namespace codee when compiled as C++ (c/stft.c:24-27).  The files of this directory carry the reference's file
 * names and hold only this: the declarations of libdsp_stft_hip.so (include/dsp_stft.h), inside the same namespace,
 * with C linkage.  Compiling the reference's tools with -I<this directory> -I<repo>/include instead of -Ic and
 * linking -ldsp_stft_hip instead of -lfftw3f -lfftw3 sends their calls (codee::stft_s(...) at src/stft.cpp:219) to
 * the GPU library; the tools' sources stay untouched.  The header's include guard makes every file after the first
 * a no-op, so a tool may include several of them. */
#ifndef DSP_FORWARD_H
#define DSP_FORWARD_H
#include <stddef.h>
#ifdef __cplusplus
namespace codee {
#endif
#include "dsp_stft.h"
#ifdef __cplusplus
}
#endif
#endif
