/* stands in for the reference file of this name: see dsp_forward.h */
#include "dsp_forward.h"
