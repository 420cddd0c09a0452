/* forwarding header: programs written for the reference encoder include "dsp_filters.h" */
#include "../avdsp_encoder.h"
