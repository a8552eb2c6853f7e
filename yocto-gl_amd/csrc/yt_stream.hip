// yt_stream.hip — the bit-exact build of the streaming scheduler's kernels (yt_stream.h) and of the launch of one generation
// (yt_stream_unit.h, which yt_fast.hip and yt_owntree.hip instantiate once more each for their modes).
#define YT_STREAM_KERNELS 1
#include "yt_stream_unit.h"
