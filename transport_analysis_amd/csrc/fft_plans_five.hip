#define TA_PLAN_LIST TA_PLANS_FIVE
#define TA_PLAN_FN plans_five
#include "fft_plans.inc"
