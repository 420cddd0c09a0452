/* forwarding header: programs written for the reference encoder include "dsp_header.h" */
#include "../avdsp_encoder.h"
