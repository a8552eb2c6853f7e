//
// ytrace_hip_prelude.h — builds the reference's UNMODIFIED apps/ytrace.cpp on the
// MI355X back-end:
//
//     g++ -include yocto-gl_amd/host/ytrace_hip_prelude.h $REF/apps/ytrace.cpp ...
//
// The prelude first includes every header the app includes (their include guards
// make the app's own #includes no-ops), then redirects the six calls the app makes
// into yocto_trace.h's lower-level API (apps/ytrace.cpp:124-158, 164-216) to their
// same-signature twins in namespace yocto::hip.  Nothing else of the app changes:
// command line, scene loading, sampler fallback, progress output, image saving are
// the reference's own code.  (In a yocto-gl checkout the same effect is the
// three-line `if (params.hipbackend)` branch shown in INTEGRATION.md.)
//
// trace_samples maps to the resident variant: the progressive loop never looks at
// the state's host arrays, and get_image() fetches only the image from the device.
//
#ifndef YTRACE_HIP_PRELUDE_H
#define YTRACE_HIP_PRELUDE_H

#include <yocto/yocto_cli.h>
#include <yocto/yocto_gui.h>
#include <yocto/yocto_math.h>
#include <yocto/yocto_scene.h>
#include <yocto/yocto_sceneio.h>
#include <yocto/yocto_shape.h>
#include <yocto/yocto_trace.h>

#include "yocto_hiptrace.h"

#define make_trace_bvh hip::make_trace_bvh
#define make_trace_lights hip::make_trace_lights
#define make_trace_state hip::make_trace_state
#define trace_samples hip::trace_samples_resident
#define get_image hip::get_image
#define trace_start hip::trace_start
#define trace_cancel hip::trace_cancel
#define trace_preview hip::trace_preview

#endif
