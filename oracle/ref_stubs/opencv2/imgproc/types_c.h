#include "ll_stub_cv.h"
