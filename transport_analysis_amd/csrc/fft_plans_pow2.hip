#define TA_PLAN_LIST TA_PLANS_POW2
#define TA_PLAN_FN plans_pow2
#include "fft_plans.inc"
