/* zxc.h — umbrella header (reference include/zxc.h). */
#ifndef ZXC_H
#define ZXC_H
#include "zxc_constants.h"
#include "zxc_error.h"
#include "zxc_opts.h"
#include "zxc_buffer.h"
#include "zxc_seekable.h"
#include "zxc_dict.h"
#include "zxc_pstream.h"
#include "zxc_mi355x.h"
#endif
