#include "ll_stub_misc.h"
