/* forwarding header: programs written for the reference encoder include "dsp_encoder.h" */
#include "../avdsp_encoder.h"
