/* forwarding header: programs written for the reference encoder include "dsp_fileaccess.h" */
#include "../avdsp_encoder.h"
