/* placeholder translation unit: single-frame fit restatement is added in a later milestone */
#include "esl_oracle.h"
