#include "ll_stub_pcl.h"
