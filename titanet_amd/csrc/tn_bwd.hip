// placeholder until the backward kernels land
#include "tn_internal.h"
int plan_backward(tn_plan*, float, const float*, const float*, float*, hipStream_t) { return TN_E_UNSUPPORTED; }
