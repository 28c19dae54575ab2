/*
 * pwnhip.h -- C ABI of libpwnhip.so, the MI355X (gfx950) implementation of
 * pwnfps's per-pixel portal ray-march path.
 *
 * Plain C: opaque context, plain pointers and sizes, int return codes
 * (0 = ok, negative = PWN_E*).  Nothing here aborts or asserts.
 * Each entry cites the reference interface it replaces (paths relative to
 * the reference tree).  INTEGRATION.md shows the host-side change.
 *
 * Like the reference's render path (globals, one main thread: main.c:26-34) a
 * context is not re-entrant: one thread at a time per context.  Trace launches
 * of a context take turns on four sets of work-queue counters: launch n counts in
 * set n mod 4 and clears the set of launch n + 2.  So a caller of the strip forms
 * either keeps all its launches on ONE stream, or alternates strictly between
 * TWO (launch n on stream n mod 2, what the frames in flight and the row tiling
 * do themselves): launch n + 2 is then ordered behind launch n, and launch n + 1
 * may run beside it.  Any other pattern needs the caller's own ordering (events)
 * between launches two apart.  The blocking frame call is ordered behind the
 * frames in flight.  Different contexts (one per GPU, one process per GPU) are
 * independent.
 */
#ifndef PWNHIP_H
#define PWNHIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PWN_OK            0
#define PWN_EINVAL       -1  /* bad argument (NULL, size, row range, w%4 with blur) */
#define PWN_ENODEV       -2  /* no usable HIP device / not gfx950 */
#define PWN_ENOMEM       -3  /* host or device allocation failed */
#define PWN_EIO          -4  /* level file could not be read (level.h:110-115) */
#define PWN_EHIP         -5  /* a HIP runtime call failed; see pwn_last_error() */
#define PWN_ENOLEVEL     -6  /* render called before a level was uploaded */
#define PWN_ETOOBIG      -7  /* sphere tables exceed the on-chip (LDS) budget */
#define PWN_EBUSY        -8  /* the frame slot is still in flight (pwn_wait_frame it first) */
#define PWN_ENOTSUP      -9  /* not available here (RCCL could not be loaded; not configured) */
#define PWN_ETIMEDOUT    -10 /* the row tiling waited longer than its deadline for a peer (pwn_tiled_set_timeouts); the tiling is
                                dead from then on (its communicator was aborted): pwn_tiled_shutdown, or leave the process */

typedef struct pwn_ctx pwn_ctx;

/* reference `portal` (defs.h:87-94); double1/double2 are never read */
typedef struct pwn_portal { int32_t x1, z1, x2, z2, rot12, c1, c2; } pwn_portal;

/* the arguments of obj_set(...,"sphere",...) (script.h:20-32), i.e. the live
   fields of `part.sph` (defs.h:73-79): radius, reflectivity, centre, colour b,g,r */
typedef struct pwn_sphere { float r, refl, x, y, z, cb, cg, cr; } pwn_sphere;

/* work counters of the last counted frame + device timings of the last frame */
typedef struct pwn_stats
{
	uint64_t rays;          /* trace_ray invocations            (trace.h:186) */
	uint64_t steps;         /* iterations of the cell walk      (trace.h:250) */
	uint64_t portals;       /* portal crossings                 (trace.h:576) */
	uint64_t sphere_tests;  /* sphere candidates tested         (trace.h:253) */
	uint64_t exhausted;     /* rays that ran out of maxsteps    (trace.h:677) */
	uint64_t wave_steps;    /* walk-loop iterations summed over wave64s: steps / (64 *
	                           wave_steps) = active-lane fraction of the walk loop   */
	float trace_ms;         /* trace kernel, HIP events                        */
	float blur_ms;          /* blur kernel(s), HIP events                      */
	float total_ms;         /* whole pwn_trace_screen_centred call incl. D2H   */
	float reserved_;
	/* divergence profile of the walk loop: how many wave64 iterations entered each code
	   path with at least one lane.  0 sphere list of the cell, 1 room body, 2 fog,
	   3 two-level transition (" / # / &), 4 ramp, 5 portal, 6 solid, 7 sphere hit maths */
	uint64_t wave_paths[8];
	/* PWN_SCHED_REFILL: passes of a wave64 through the shade / new-pixel / set-up phase, and the
	   lanes that had a ray to shade in them (phase_lanes / (64 * phase_passes) = their lane use) */
	uint64_t phase_passes, phase_lanes;
	/* PWN_OPT_WAVE_LOG: residency of the trace kernel's wave64s in the last frame, in ticks of the GPU's
	   constant 100 MHz clock: sum of the waves' lifetimes, first start to last end, number of waves.
	   wave_time / (waves * kernel_span) = share of the kernel's duration the average wave was resident */
	uint64_t wave_time, kernel_span, waves;
	/* counted frames: how often a wave64 entered each region of the kernel outside the walk's paths above (the issue
	   model, tools/issue_model.py; the order is trace_common.h's RG_* enum): 0 ray segment set-up, 1 its slow path, 2 exhausted
	   rays, 3 wall shading, 4 sphere shading, 5 floor normal, 6 sphere mirror, 7 jitter + push, 8 / 9 composite of the last
	   bounce and its fog, 10 / 11 of the first, 12 help-another-queue, 13 units, 14 sphere tests (list loop trips), 15
	   nearer-sphere updates, 16 walk iterations with a lane in a non-room cell, 17 units in the right half of a 32-pixel tile,
	   18 / 19 height-change block: out of a 2-high cell, and its wall test, 20 portal letters that are walls, 21 portal
	   crossings, 22 / 23 those that turn by a quarter / by a half, 24 waves of the launch */
	uint64_t regions[32];
} pwn_stats;

/* options for pwn_set_option */
#define PWN_OPT_BLUR_PASSES 1  /* POSTPROC_BLUR (defs.h:9); default 1, 0 = off */
#define PWN_OPT_COUNTERS    2  /* 1: frames also fill pwn_stats counters (slower) */
#define PWN_OPT_SCHEDULER   3  /* how the trace kernel hands pixels to the lanes of a wave64: */
#define PWN_SCHED_UNITS     0  /*   a wave traces 16x4-pixel units, all 64 lanes in step (ray set-up, walk, shading) */
#define PWN_SCHED_REFILL    1  /*   lanes whose ray ended are refilled by ballot + prefix rank while the rest walk on */
#define PWN_SCHED_DEFAULT   PWN_SCHED_UNITS
#define PWN_OPT_REFILL_LIMIT 4 /* PWN_SCHED_REFILL: lane-steps (1..64000) the ended rays of a batch may wait, in sum, for
                                  the batch's other rays before they are shaded and replaced; rays still walking
                                  then walk on beside the next batches */
#define PWN_REFILL_LIMIT_DEFAULT 256
#define PWN_OPT_WAVE_LOG     6 /* 1: every wave64 of the trace kernel stamps its start and end (two clock reads and one
                                  store per wave); pwn_get_stats then fills wave_time / kernel_span / waves */
#define PWN_OPT_FRAME_OVERLAP 7 /* frames in flight: 1 (default) = the kernels of successive frames alternate between two compute
                                  streams, so that the next frame's trace grid fills the CUs the current one's tail leaves idle
                                  and its blur runs beside it; 0 = all frames on one stream, strictly one after the other.
                                  PWN_EBUSY while frames are in flight.  A slot's frame is complete when its pwn_wait_frame
                                  returns (every slot has its own event); frames two apart complete in order, neighbours may
                                  finish either way round.  Frames f and f + 1 share nothing they write: pre-blur planes by parity,
                                  colour / depth / surface planes by slot, counter sets by launch, and the tables a frame reads are
                                  the copy that was current at its submit (four copies; an upload never touches one in use).
                                  A frame counted (PWN_OPT_COUNTERS) or wave-logged runs on one stream: one set of counters.
                                  The second stream is created at another priority level than the first so that it gets a
                                  hardware queue of its own.  The row tiling reads the option at pwn_tiled_init. */
#define PWN_OPT_TRACE_ROOM 8   /* frames on two compute streams (PWN_OPT_FRAME_OVERLAP, and the row tiling): workgroups the persistent
                                  trace grid leaves free so that the other stream's kernels find room on every CU beside it
                                  instead of waiting for its end.  -1 (default): the library measures -- windows of delivered
                                  frames with no room and with one workgroup per CU, the better kept for ~500 frames (half a second
                                  at least), then again (the answer depends on the scene: +3.5 % at 4K on level.txt, -3 % on a hall of
                                  mirrors); >= 0: that many, always.  Results never depend on it. */
#define PWN_OPT_TILED_CHOREO 10 /* row tiling, read by pwn_tiled_init (PWN_EBUSY while a tiling exists; PWN_TILED_CHOREO=split in the environment
                                  for a process).  PWN_TILED_CHOREO_INSTREAM (default): pwn_tiled_submit(f) enqueues everything of frame f --
                                  trace, halo rows, blur, gather, words -- in order on the frame's own compute stream (frames alternate between
                                  two); frame f's exchange overlaps the other stream's trace f+1 and no event crosses a queue.
                                  PWN_TILED_CHOREO_SPLIT (rounds 2-3): the exchanges on a third stream tied to the kernels by four events per
                                  frame, blur f enqueued by submit f+1 and its gather by submit f+2.  Same frames either way.  Measured on one
                                  GPU (profiles/r4/host_bound.txt): split costs a rank 0.14-0.21 ms per frame whatever the frame's
                                  size -- a strip of an 8-way tiled 4K frame needs 0.05 ms of kernels. */
#define PWN_TILED_CHOREO_INSTREAM 0
#define PWN_TILED_CHOREO_SPLIT    1
#define PWN_OPT_TILED_COMMS 12   /* row tiling over RCCL, in-stream choreography, read by pwn_tiled_init (PWN_EBUSY while a tiling exists;
                                  PWN_TILED_COMMS=perstream in the environment).  PWN_TILED_COMMS_ONE (default): one communicator for all
                                  exchanges -- it runs its grouped launches in the order they were made, so frame f+1's halo rows start
                                  behind frame f's gather although the two are on different streams.  PWN_TILED_COMMS_PER_STREAM: one
                                  communicator per compute stream (each brought up like the first, under the deadline; its id travels from
                                  rank 0 over the first): the streams' exchanges are independent.  Blocking mode only.  Never run between
                                  GPUs (nor has anything else here); on one GPU, a rank's exchange with itself: DESIGN.md 6. */
#define PWN_TILED_COMMS_ONE        1
#define PWN_TILED_COMMS_PER_STREAM 2
#define PWN_OPT_TILED_STREAMS 11 /* row tiling, in-stream choreography, read by pwn_tiled_init (PWN_EBUSY while a tiling exists; PWN_TILED_STREAMS=n
                                  in the environment): 2 = frames alternate between the context's two compute streams, 3 = they rotate
                                  over those and a third that the tiling creates -- two frames' exchanges and blurs then overlap a third
                                  frame's trace.  Same frames.  Without PWN_OPT_FRAME_OVERLAP there is one stream whatever this says. */
#define PWN_OPT_UNIT_ORDER 9   /* 0 (default): the trace kernel hands its 16 x 4-pixel units out in arithmetic order, rows from the frame's
                                  middle row outwards.  1 (PWN_UNIT_ORDER=1 in the environment for a process): dearest first -- every unit's cost
                                  (the time its wave spent on it) is written by the launch, put in order per work queue behind the frame's last
                                  kernel (a counting sort, ~3 us), and used by the next launch of the same rows on that stream, so that the units
                                  handed out last are the cheap ones.  Measured (profiles/r4/unit_order_ab.txt): a launch of a few units per wave
                                  that runs alone gets shorter (720p -9 %, a middle strip of an 8-way 4K tiling -10 %), long launches and frames
                                  on two streams get slower (4K +2 ... +4 %): sorted by cost, units of one kind share a SIMD.  Hence off.  The
                                  first launch of a geometry uses the arithmetic order; with the strip forms the sort rides behind
                                  pwn_blur_rows_device[_bounded] on the caller's stream.  Any order gives the same pixels.  The reference's
                                  counterpart is OpenMP's static schedule over 32-row chunks (screen.h:63-64). */
#define PWN_OPT_CALL_STRIPS 13 /* the blocking call, pwn_trace_screen_centred (the one call an unchanged reference loop makes per frame,
                                  main.c:107): handing a 4K frame to the host over PCIe takes twice as long as its kernels, so the call
                                  works in row strips -- strip k is traced while strip k - 1 is blurred from the rows traced so far and
                                  strip k - 2 travels to sbuf / zbuf on the copy stream.  A tap of the blur that lands below the rows
                                  traced so far (taps reach 0.002*h*|depth-1| rows, screen.h:100-102; a strip is blurred where taps of
                                  depth <= 8 are covered) is counted; such a frame's blur is repeated over the whole frame before the call
                                  returns and the next calls cover depth 24; a frame that exceeds that too sends the next calls through in
                                  one piece for a while: the frame delivered is always the exact one.  -1 (default): strips for frames of 3 Mpixels
                                  and more into registered buffers (pwn_host_register), of 6 Mpixels and more into others, with
                                  POSTPROC_BLUR 0 or 1, not for counted or wave-logged frames; 0: one launch per pass,
                                  then the copies (what pwn_stats.trace_ms / blur_ms time as single launches); 2..32: that many
                                  strips whatever the size.  With strips pwn_stats.trace_ms runs from the first strip's trace to the
                                  end of the last one's (the blurs between them included), blur_ms is the tail behind it. */
#define PWN_OPT_FRAME_TIMING 5 /* frames in flight: record HIP events around the trace and blur kernels of every N-th
                                  frame (pwn_frame.timed, .trace_ms ...); 1 = every frame (default), 0 = never.  An event
                                  between two kernels costs a few microseconds of pipeline */

/*
 * Replaces the buffer/global set-up of main.c:26-34,395-400 (rwidth, rheight,
 * sbuf, tsbuf, zbuf).  `device` is the HIP ordinal.  Device depth is
 * zero-initialised; like the reference's zbuf it keeps its previous value at
 * pixels whose primary ray exhausts maxsteps (trace.h:677).
 */
int pwn_init(pwn_ctx **out, int device, int width, int height);
void pwn_destroy(pwn_ctx *ctx);
/*
 * The same for several GPUs of ONE process.  The reference's host is one process with one loop (main.c:93-109) whose row
 * parallelism -- the OpenMP pragmas of screen.h:63-67,77 -- is invisible to it; so is this: the handle is used exactly
 * like pwn_init's (the level and object calls, pwn_trace_screen_centred, the frames in flight, pwn_screen_upscale,
 * pwn_get_stats, pwn_host_register), and every frame is row-tiled over the devices behind it -- the choreography of
 * pwn_tiled_* (strip per device, halo rows to the neighbours, bounded blur with exact repeat, moving cuts), driven by one
 * library-owned thread per device, the devices' tables filled from the one level and object table of the handle.  No
 * unique id, no second process, no replicated game state.
 *   devices[ndev]   HIP ordinals, 1..PWN_TILED_MAX_WORLD of them.  The same ordinal may appear more than once: so many
 *                   members share that device (tests on a box with one GPU; no use otherwise).  ndev = 1 is pwn_init.
 *   frames          pwn_trace_screen_centred: every device copies its finished strip (and its depth strip) straight into the
 *                   caller's sbuf / zbuf over its own PCIe link.  pwn_frames_config with PWN_FRAME_SBUF / _ZBUF: the same
 *                   into the group's pinned frames, up to PWN_MAX_SLOTS in flight; PWN_FRAME_SURFACE: every device upscales its own
 *                   strip (screen_upscale, screen.h:126-149) and copies those rows of the surface to the host as well; with flags 0
 *                   the finished strips are gathered on devices[0] (pwn_frame.d_sbuf) and nothing goes to the host.
 *   exchange        between devices: RCCL (ncclSend / ncclRecv groups over xGMI, one communicator rank per device, made
 *                   in-process under the bring-up deadline) when the ordinals are distinct and librccl loads, else -- and
 *                   with PWN_GROUP_TRANSPORT=local in the environment -- PWN_TRANSPORT_LOCAL: peer-to-peer copies behind
 *                   events.  pwn_group_info says which.
 *   errors          a member's error is the call's (pwn_last_error names the member); a member that stops answering ends the
 *                   call with PWN_ETIMEDOUT after pwn_tiled_set_timeouts' limits, which the handle takes too.
 * pwn_tiled_* (other than set_timeouts) and the strip forms are refused on such a handle (PWN_ENOTSUP): it runs its own tiling.
 */
typedef struct pwn_group_info
{
	int members, transport;                  /* PWN_TRANSPORT_RCCL or _LOCAL */
	int devices[64];                         /* the ordinals as given (PWN_TILED_MAX_WORLD) */
	int cuts[65];                            /* member m traces rows [cuts[m], cuts[m + 1]) of the next frame */
	int halo_rows, host_sink;                /* of the tiling in force (0, 0 before the first frame) */
	uint64_t frames, frames_redone, recuts;  /* delivered; repeated with whole strips; how often the cuts moved */
	char note[160];                          /* why the transport is not RCCL (members on one device; librccl did not load; its
	                                            bring-up between the members failed -- the group then goes on with copies, once, unless
	                                            PWN_GROUP_TRANSPORT named the transport), or empty */
} pwn_group_info;
int pwn_init_multi(pwn_ctx **out, const int *devices, int ndev, int width, int height);
int pwn_group_info_get(pwn_ctx *ctx, pwn_group_info *out);
int pwn_set_option(pwn_ctx *ctx, int option, int value);
/* PWN_OPT_UNIT_ORDER as it stands: out[0] the option, out[1] trace launches so far that handed their units out in a sorted order,
   out[2] sorts launched (one behind every frame whose trace wrote its units' costs), out[3] units the first compute stream's
   current order covers (0: none yet). */
int pwn_unit_order_state(pwn_ctx *ctx, unsigned long long out[4]);
/* the sort by itself (tests): `units` costs in (host memory), the order out -- perm_out holds 64 * cap entries, cap = ceil(units / 64);
   queue q's units, dearest first, are perm_out[q * cap .. q * cap + its length), the rest of its row is left 0xffffffff */
int pwn_unit_order_probe(pwn_ctx *ctx, const uint16_t *cost, uint32_t units, uint32_t *perm_out);
/* Trace launches of a context take their work-queue counters from sets used in turn, and launch n resets the set of launch n - R
   (R = 2; 3 for a tiling on three compute streams): launches R apart have to be ordered.  On one stream, and with frames rotating over
   R streams, they are by themselves; a launch that leaves the rotation (pwn_trace_screen_centred behind frames in flight, a counted
   frame) is put behind the launch R before it with an event.  *out = how often that happened (tests). */
int pwn_launch_order_waits(pwn_ctx *ctx, unsigned long long *out);
/* PWN_OPT_TRACE_ROOM as it stands: out[0] the option's value (-1 = measuring), out[1] the workgroups the next two-stream trace launch
   leaves free, out[2] how often the two settings were compared, out[3] how often the setting changed */
int pwn_trace_room_state(pwn_ctx *ctx, int out[4]);
const char *pwn_strerror(int code);
const char *pwn_last_error(pwn_ctx *ctx);

/* level_load (level.h:107-228): parse the ASCII grid (CR/LF rules, '*' spawn,
   lower-case portal quirk), build pmap, upload.  */
int pwn_level_load(pwn_ctx *ctx, const char *path);
int pwn_level_load_mem(pwn_ctx *ctx, const char *text, int len);
/* the same tables handed over ready-made: lv->data / lv->pmap (defs.h:103-105) */
int pwn_upload_level(pwn_ctx *ctx, const uint8_t data[4096], const pwn_portal pmap[26]);
int pwn_get_level(pwn_ctx *ctx, uint8_t data[4096], pwn_portal pmap[26], int32_t spawn[2]);

/* level_prepare_render (level.h:64-81) + level_part_add_bbox (level.h:1-19):
   bins the live spheres per cell in object order and uploads the lists */
int pwn_upload_spheres(pwn_ctx *ctx, const pwn_sphere *spheres, int n);
int pwn_get_bins(pwn_ctx *ctx, uint16_t counts[4096], int32_t *idx, int cap);

/*
 * The object table behind those lists, driven the way game.lua drives the
 * reference (script.h:1-64): lv->objs[OBJ_MAX] (defs.h:4,98-99).
 *   pwn_obj_new        level_obj_new (level.h:41-62): first freed slot, else
 *                      append; returns the object's index, PWN_ENOMEM when all
 *                      PWN_OBJ_MAX slots are taken (the reference returns NULL)
 *   pwn_obj_set_sphere obj_set(o, "sphere", r, refl, x, y, z, b, g, r)
 *                      (script.h:10-40); Lua numbers are doubles and are
 *                      narrowed to float on store, as there
 *   pwn_obj_free       obj_free (script.h:42-51): the slot becomes reusable.  As with
 *                      the reference's part pointers the handle stays usable:
 *                      obj_set on a freed slot makes it a sphere again (script.h:24
 *                      sets pt->typ whatever it was), obj_free on it changes nothing
 *   pwn_level_get      level_get(cx, cz) (script.h:53-64): the cell character
 *                      under get_cell's clamp (util.h:151-158)
 *   pwn_prepare_render level_prepare_render (level.h:64-81): bin every object
 *                      that is not free, in table order, and upload; an object
 *                      that was created but never set is PWN_EINVAL (the
 *                      reference aborts, level.h:34-37)
 *   pwn_get_objects    the live spheres in table order (at most cap); returns
 *                      their number
 * pwn_upload_spheres replaces the whole table by n set spheres.
 */
#define PWN_OBJ_MAX 10000
int pwn_obj_new(pwn_ctx *ctx);
int pwn_obj_set_sphere(pwn_ctx *ctx, int obj, double r, double refl, double x, double y, double z,
	double cb, double cg, double cr);
int pwn_obj_free(pwn_ctx *ctx, int obj);
int pwn_level_get(pwn_ctx *ctx, int cx, int cz);
int pwn_prepare_render(pwn_ctx *ctx);
int pwn_get_objects(pwn_ctx *ctx, pwn_sphere *out, int cap);

/*
 * trace_screen_centred(lv, 0, 0, rwidth, rheight, &cam) (screen.h:31-124,
 * called at main.c:107) with sec_current (defs.h:23) passed explicitly.
 * cam = mat4 rows x,y,z,w (defs.h:46-52).  Blocking.  Fills the caller's
 * host sbuf (BGRA8, pitch = width) and, if not NULL, zbuf -- the layouts of
 * main.c:31,33.
 */
int pwn_trace_screen_centred(pwn_ctx *ctx, const float cam[16], float sec_current,
	uint32_t *sbuf, float *zbuf);
/*
 * The host's frame buffers are malloc'ed once and live as long as the process (main.c:395-400).  Made known to the
 * device once (hipHostRegister), the copies of the blocking call into them are plain DMA that runs beside the kernels
 * of the frame's later strips and never holds the calling thread; into memory that is not registered every strip's
 * copy is staged by the runtime and blocks the caller while it runs (the call then keeps two strips of kernels
 * enqueued ahead of the copy it waits in).  Same pixels either way.
 *   pwn_host_register    [base, base + bytes): once, right after the malloc; PWN_EBUSY when PWN_HOST_REGS_MAX (16)
 *                        ranges are registered, PWN_EHIP when the runtime refuses (the range stays usable, unregistered)
 *   pwn_host_unregister  before the memory is freed -- a registered range that is freed and handed out again by
 *                        malloc is still the OLD pages to the device; pwn_destroy unregisters what is left
 * *_state: out[0] the option, out[1] strips of the last blocking call (1 = it ran in one piece), out[2] blocking calls
 * that ran in strips, out[3] how many of those had their blur repeated over the whole frame, out[4] the copy streams the
 * chunks go out on (1 or 2; 0 = the context is still finding out: its first 8 calls in strips use one, the next 8 two, the
 * faster stays -- which it is depends on the queues the runtime handed out; PWN_CALL_COPY_STREAMS=1|2 in the environment
 * fixes it), out[5] the depth whose taps a chunk's blur waits for (8, or 24 after a frame whose taps went further).
 */
int pwn_host_register(pwn_ctx *ctx, void *base, size_t bytes);
int pwn_host_unregister(pwn_ctx *ctx, void *base);
int pwn_call_strips_state(pwn_ctx *ctx, unsigned long long out[6]);

/*
 * Frames in flight.  The reference presents every frame on the host
 * (trace_screen_centred fills sbuf, screen_upscale fills screen->pixels, SDL_Flip:
 * main.c:107-109).  Over PCIe that hand-over takes longer than the kernels of a
 * frame, so a host that wants throughput overlaps it: with 2-3 slots, frame i
 * travels to the library's pinned host buffers on a copy stream while the
 * kernels of frame i+1 run.
 *   pwn_frames_config  nslots (1..PWN_MAX_SLOTS; 0 releases them) and what a frame
 *                      delivers to the host (flags 0: nothing, the frame stays on the
 *                      device, pwn_frame.d_*): PWN_FRAME_SBUF the colour plane (main.c:31),
 *                      PWN_FRAME_ZBUF the depth plane (main.c:33), PWN_FRAME_SURFACE
 *                      the scale x scale upscaled surface (screen_upscale,
 *                      screen.h:126-149; pitch_bytes 0 = width*scale*4; bytes between
 *                      the rows of a wider pitch read 0).  Buffers are library-owned
 *                      pinned host memory.
 *   pwn_submit_frame   enqueue trace + blur (+ sink) + the copies for `slot`; returns at
 *                      once.  Uses the tables of the last pwn_upload_spheres /
 *                      pwn_prepare_render / level call, which may be called between
 *                      submits without waiting for anything.  PWN_EBUSY if the slot's
 *                      previous frame has not been waited for.  (A host that is short of
 *                      microseconds prepares the next frame's tables BEFORE it waits for the
 *                      slot it wants to reuse: their upload runs on its own stream during that
 *                      wait and the trace launch then queues no wait for it -- ~5 us per frame.)
 *   pwn_wait_frame     block until the slot's frame is on the host; the pointers in
 *                      *out stay valid until the next submit on that slot.
 *   pwn_frame_ready    1 / 0 without blocking.
 * Frames complete in submission order.  The blocking pwn_trace_screen_centred may be
 * mixed in; it is ordered behind the submitted frames.
 */
#define PWN_MAX_SLOTS     4
#define PWN_FRAME_SBUF    1
#define PWN_FRAME_ZBUF    2
#define PWN_FRAME_SURFACE 4
typedef struct pwn_frame
{
	const uint32_t *sbuf;        /* BGRA8, pitch = width          (NULL unless PWN_FRAME_SBUF)    */
	const float *zbuf;           /* fp32 depth, pitch = width     (NULL unless PWN_FRAME_ZBUF)    */
	const uint32_t *surface;     /* upscaled, surface_pitch_bytes (NULL unless PWN_FRAME_SURFACE) */
	int surface_pitch_bytes;
	const void *d_sbuf, *d_zbuf, *d_surface;   /* the same planes on the device (d_surface NULL without the flag) */
	float sec_current;           /* as submitted */
	float trace_ms, blur_ms, sink_ms;   /* device time of this frame's kernels, if timed (PWN_OPT_FRAME_TIMING) */
	int timed;
	uint64_t seq;                /* 1, 2, ... in submission order */
} pwn_frame;
int pwn_frames_config(pwn_ctx *ctx, int nslots, int flags, int scale, int pitch_bytes);
int pwn_submit_frame(pwn_ctx *ctx, const float cam[16], float sec_current, int slot);
int pwn_wait_frame(pwn_ctx *ctx, int slot, pwn_frame *out);
int pwn_frame_ready(pwn_ctx *ctx, int slot);
/* blocking copy of a device plane of a waited-for frame (pwn_frame.d_*) to host memory, for hosts
   that keep their frames on the device and look at one now and then */
int pwn_read_plane(pwn_ctx *ctx, const void *d_src, void *dst, size_t bytes);

/*
 * Strip forms: the two passes restricted to a band of rows, for callers that run their own
 * choreography (the library's own row tiling, pwn_tiled_* below, is built from the same two
 * launches and does the exchange between the GPUs itself: RCCL send / recv inside the library).
 * Pointers are DEVICE pointers to FULL frames (pitch = width); only rows
 * [y0,y1) are written.  Stream-ordered on `stream` (a hipStream_t, NULL =
 * default stream); they do not synchronise.
 *   trace:  the OpenMP loop of screen.h:61-67 restricted to rows [y0,y1)
 *   blur:   one pass of screen.h:77-121 restricted to rows [y0,y1); reads
 *           the whole pre-blur frame (the role of tsbuf) and depth rows [y0,y1)
 */
int pwn_trace_rows_device(pwn_ctx *ctx, const float cam[16], float sec_current,
	int y0, int y1, void *d_sbuf, void *d_zbuf, void *stream);
int pwn_blur_rows_device(pwn_ctx *ctx, int y0, int y1, const void *d_pre,
	const void *d_zbuf, void *d_out, void *stream);
/*
 * The same pass when the caller exchanged only a bounded halo instead of the
 * whole pre-blur frame: rows [avail_y0, avail_y1) of d_pre hold this frame.
 * A tap outside them (taps reach 0.002*h*|depth-1| rows, screen.h:100-102)
 * adds to the uint32 at d_miss (device memory, caller clears it); the strip
 * is then not valid and has to be repeated with the whole frame present.
 */
int pwn_blur_rows_device_bounded(pwn_ctx *ctx, int y0, int y1, const void *d_pre,
	const void *d_zbuf, void *d_out, int avail_y0, int avail_y1, void *d_miss, void *stream);

/*
 * Row tiling behind the ABI: the choreography of the strip forms with the exchange done by
 * the library -- RCCL over xGMI, two grouped ncclSend/ncclRecv launches per frame.  One
 * process per GPU, each with its own context of the FULL frame size; every rank makes the
 * same calls in the same order (level and sphere uploads included: tables are per rank).
 *   pwn_tiled_unique_id  rank 0: the id of the group (an ncclUniqueId for PWN_TRANSPORT_RCCL);
 *                        the host hands its 128 bytes to the other ranks (a file, a pipe, MPI ...)
 *   pwn_tiled_init       collective.  Rank r owns rows [r*per, (r+1)*per), per = ceil(h/world)
 *                        rounded up to 8 (the OpenMP loops of screen.h:63,77 split by rows) -- to begin with: the cuts move
 *                        with what the rows cost (pwn_tiled_balance below).
 *                        halo_rows: pre-blur rows exchanged with each neighbour strip; < 0 = default
 *                        (depth 24: 0.002*h*24 + 2 rows, screen.h:100-102), 0 = every strip to
 *                        everybody (an all-gather).  The blur counts taps that leave the halo; a frame
 *                        with such a tap is repeated with whole strips before it is delivered, and
 *                        whole strips are used from then on: delivered frames are always exact.
 *                        POSTPROC_BLUR 0 or 1.
 *   pwn_tiled_submit     enqueue one frame on every rank, in order on the frame's compute stream: trace own strip,
 *                        one grouped exchange (the strip's border rows with the neighbour strips), blur the strip,
 *                        a second grouped exchange (the finished strip to the frame's root, the rank's two words to
 *                        every rank).  Frames alternate between two streams, so that a frame's exchange overlaps the
 *                        next frame's trace: a frame then costs max(kernels, exchange), not their sum
 *                        (PWN_OPT_TILED_CHOREO).  At most PWN_TILED_SLOTS - 1 = five frames in flight (PWN_EBUSY).
 *   pwn_tiled_wait       every rank: block until the oldest frame in flight is complete; on rank 0 (the frame's root)
 *                        out->d_sbuf is the full frame on the device (valid until five more frames
 *                        were submitted) and, with PWN_TILED_HOST, out->sbuf a pinned host copy.
 *   pwn_tiled_host_sink  optional, every rank, after pwn_tiled_init and before the first frame: frames are
 *                        delivered to the HOST instead (main.c:107-109 presents every frame there).  `base` is
 *                        host memory for PWN_TILED_SLOTS whole frames (w*h*4 bytes each, pitch = width), the SAME
 *                        memory in every rank -- POSIX shared memory mapped by every process -- which the library
 *                        registers with its device.  Every rank then copies its finished strip straight into the
 *                        frame over its own PCIe link (N links, not rank 0's one) and there is no gather to rank
 *                        0: the grouped exchange carries the halo rows and one word per rank, sent behind that
 *                        rank's copy, so that a frame is delivered when every strip has landed.  pwn_tiled_wait
 *                        then gives out->sbuf on EVERY rank (valid until the second next pwn_tiled_submit),
 *                        out->d_sbuf is NULL.
 *   pwn_tiled_gather_root  optional, collective (the same call on every rank, no frame in flight): which rank a frame is
 *                        gathered on.  PWN_TILED_ROOT_FIXED (default): rank 0, every frame -- its links then carry (N-1)/N
 *                        of every frame, which bounds the frame rate of a tiling whose kernels are faster than that
 *                        (xGMI is point to point: 7 links into one GPU).  PWN_TILED_ROOT_ROTATE: frame f
 *                        (pwn_tiled_frame.seq - 1) is gathered on rank f mod N, for consumers that sit on every GPU
 *                        (an encoder per device, N displays): every rank then takes in 1/N of the frames and each of
 *                        its links carries one strip every N-th frame.  pwn_tiled_wait gives out->d_sbuf (and, with
 *                        PWN_TILED_HOST, out->sbuf) on the frame's root, out->root says which rank that is.
 *   pwn_tiled_shutdown   collective; pwn_destroy does it too.
 *   pwn_tiled_set_timeouts  this rank, any time (also before pwn_tiled_init): how long a call of the tiling may wait for
 *                        the other ranks before it gives up.  The reference's error model is print-and-return
 *                        (level.h:35-37,110-115); a row tiling adds a failure the reference cannot have -- a peer that never
 *                        arrives -- and that must come back as an error too, not as a hang.  init_ms bounds the bring-up
 *                        of pwn_tiled_init (communicator + one word exchanged with EVERY other rank, so that every
 *                        connection a frame will use exists when it returns), wait_ms bounds pwn_tiled_wait (and what
 *                        pwn_tiled_submit may have to wait for inside the transport).  On expiry the RCCL communicator
 *                        is aborted (ncclCommAbort; the kernels it has on the device leave), the call returns
 *                        PWN_ETIMEDOUT with pwn_last_error() naming this rank, what it waited for and how far it got,
 *                        and every later pwn_tiled_submit / _wait returns PWN_ETIMEDOUT at once.  0 keeps a value, < 0
 *                        restores the default (PWN_TILED_INIT_TIMEOUT_MS / PWN_TILED_WAIT_TIMEOUT_MS in the environment,
 *                        else 120 s / 60 s).
 *   pwn_tiled_preflight  this rank, no tiling needed: what a first multi-GPU run wants to know before it starts, as one
 *                        JSON object in `json` -- the devices this process sees and, from the context's device, which of
 *                        them it can reach directly (hipDeviceCanAccessPeer), the librccl that dlopen resolved (path and
 *                        version), how the communicator will be driven (PWN_TILED_RCCL_MODE) and the deadlines.  Returns
 *                        the length written (the text is cut at n - 1), or PWN_E*.
 * PWN_TRANSPORT_SHM moves the same messages through POSIX shared memory instead: for tests
 * on a box with one GPU, where RCCL cannot run two ranks; the ranks may share a device.
 */
#define PWN_TILED_ID_BYTES 128
#define PWN_TRANSPORT_RCCL 0
#define PWN_TRANSPORT_SHM  1
#define PWN_TRANSPORT_LOCAL 2      /* between the members of a group inside one process (pwn_init_multi): device-to-device copies behind events */
#define PWN_TILED_HOST     1
#define PWN_TILED_SLOTS    6       /* frames a host sink holds (at most five in flight and the one being reused) */
#define PWN_TILED_MAX_WORLD 64
#define PWN_TILED_ROOT_FIXED  0
#define PWN_TILED_ROOT_ROTATE 1
typedef struct pwn_tiled_frame
{
	const void *d_sbuf;          /* the frame's root (rank 0 unless pwn_tiled_gather_root): the frame on the device, BGRA8, pitch = width; NULL elsewhere */
	const uint32_t *sbuf;        /* the root with PWN_TILED_HOST: pinned host copy; with a host sink: the frame, on every rank */
	uint64_t seq;                /* 1, 2, ... in submission order */
	int redone;                  /* 1: a tap left the halo and the frame was repeated with whole strips */
	int timed;                   /* PWN_OPT_FRAME_TIMING sampled this frame: */
	float trace_ms, frame_ms;    /*   this rank's trace kernel; its trace .. blur incl. waiting for the exchange */
	float blur_ms;               /*   its blur kernel */
	float halo_ms, gather_ms;    /*   the two grouped exchanges: this frame's halo rows (G1); the gather that carried this frame's
	                                  strips and words (G2; split choreography: 0 when pwn_tiled_wait had to launch it itself) */
	float enqueue_us;            /* host time inside pwn_tiled_submit for this frame (every frame) */
	int y0, y1;                  /* the rows this rank traced of this frame (the cuts move: pwn_tiled_balance) */
	uint32_t cost;               /* what they cost: sum of the trace waves' lifetimes, ticks of the GPU's 100 MHz clock */
	int root;                    /* the rank this frame was gathered on (pwn_tiled_gather_root; -1 with a host sink) */
} pwn_tiled_frame;
typedef struct pwn_tiled_info
{
	int rank, world, y0, y1, rows_per_rank, halo_rows, transport;      /* y0, y1: this rank's rows of the NEXT frame */
	uint64_t frames, frames_redone, groups;        /* delivered frames; repeated ones; grouped exchanges launched */
	uint64_t bytes_sent, bytes_received;           /* by this rank */
	uint64_t bytes_to_host;                        /* host sink: copied into the host frame by this rank */
	int host_sink;                                 /* 1: every rank delivers its strip to the host (pwn_tiled_host_sink) */
	int max_rows;                                  /* the tallest strip a rank may be given (1.5 equal strips) */
	int balance_every;                             /* pwn_tiled_balance */
	int grid_reserve;                              /* workgroups the trace grid leaves free for RCCL's kernels (pwn_tiled_set_reserve) */
	int two_streams;                               /* 1: frames alternate between two compute streams (PWN_OPT_FRAME_OVERLAP at init) */
	uint64_t recuts;                               /* how often the cuts moved */
	int gather_root;                               /* PWN_TILED_ROOT_* (pwn_tiled_gather_root) */
	int rccl_nonblocking;                          /* RCCL transport: 1 = a non-blocking communicator, every call polled against the deadline
	                                                  (PWN_TILED_RCCL_MODE=nonblocking); 0 = a blocking one, bring-up on a helper thread under
	                                                  the deadline (the default) */
	int init_timeout_ms, wait_timeout_ms;          /* pwn_tiled_set_timeouts, as in force */
	int dead;                                      /* 1: a deadline passed or the transport failed; the communicator is gone */
	int compute_streams;                           /* 1, 2 or 3: frame f runs on compute stream f mod this (PWN_OPT_FRAME_OVERLAP, PWN_OPT_TILED_STREAMS) */
	int choreography;                              /* PWN_TILED_CHOREO_* as fixed at pwn_tiled_init */
	int communicators;                             /* RCCL transport: communicators in use (PWN_OPT_TILED_COMMS); 0 over shared memory */
} pwn_tiled_info;
int pwn_tiled_unique_id(void *id128, int transport);
int pwn_tiled_init(pwn_ctx *ctx, int rank, int world, const void *id128, int transport, int halo_rows);
int pwn_tiled_submit(pwn_ctx *ctx, const float cam[16], float sec_current);
int pwn_tiled_wait(pwn_ctx *ctx, int flags, pwn_tiled_frame *out);
int pwn_tiled_host_sink(pwn_ctx *ctx, void *base, size_t bytes);
int pwn_tiled_gather_root(pwn_ctx *ctx, int mode);
int pwn_tiled_get_info(pwn_ctx *ctx, pwn_tiled_info *out);
void pwn_tiled_shutdown(pwn_ctx *ctx);
int pwn_tiled_set_timeouts(pwn_ctx *ctx, int init_ms, int wait_ms);
int pwn_tiled_preflight(pwn_ctx *ctx, char *json, size_t n);
/*
 * Moving cuts.  Equal strips are not equal work (the horizon band of level.txt costs 1.2-1.3x the mean strip of an
 * 8-way tiling), and the reference's answer to that -- OpenMP's static schedule over 32-row chunks, screen.h:63-64 --
 * is the equal split.  Here every rank's trace launch measures what its rows cost (the sum of its waves' lifetimes),
 * the number travels to every rank with the frame's miss word, and every `every_frames` delivered frames all ranks
 * compute the same new cuts from the same numbers: piecewise-linear cost over the rows, cut k where it reaches k / world
 * of the total, multiples of 8 rows, every strip at least the halo and at most pwn_tiled_info.max_rows rows.  New cuts
 * take effect with the next submitted frame; a frame in flight keeps the cuts it was traced with (exchange, blur,
 * gather, host-sink copies, and the repeat after a missed halo all use the frame's own).
 *   pwn_tiled_balance   collective (the same call on every rank, between frames): the period in delivered frames,
 *                       0 = the cuts stay where they are.  Default 8 (PWN_TILED_BALANCE=k in the environment
 *                       overrides; 0 when a strip of the equal split would be empty, or without blur).
 *   pwn_tiled_set_cuts  collective: n = world + 1 boundaries, 0 = cuts[0] < cuts[1] < ... < cuts[world] = height,
 *                       inner ones multiples of 8, strips within [halo, max_rows]; for the next submitted frame.
 *   pwn_tiled_get_cuts  the cuts of the next frame (world + 1 ints) and, if cost != NULL, every rank's cost word of
 *                       the last delivered frame (world words); returns world + 1.
 *   pwn_tiled_set_reserve  this rank, between frames: workgroups the persistent trace grid leaves free so that
 *                       RCCL's send / recv kernels find a CU to start on (default 16 with the RCCL transport, 0 = fill
 *                       every CU; PWN_TILED_RESERVE in the environment sets the default).
 */
int pwn_tiled_balance(pwn_ctx *ctx, int every_frames);
int pwn_tiled_set_cuts(pwn_ctx *ctx, const int *cuts, int n);
int pwn_tiled_get_cuts(pwn_ctx *ctx, int *cuts, uint32_t *cost);
int pwn_tiled_set_reserve(pwn_ctx *ctx, int workgroups);
/* the re-cut rule by itself (no context, no device): cuts[world + 1] and the strips' costs in, out[world + 1]; returns 1
   if the cuts moved, 0 if they stay (balanced within 2 %, a zero cost, constraints that cannot be met) */
int pwn_tiled_recut(const int *cuts, const uint32_t *cost, int world, int height, int min_rows, int max_rows, int *out);

/* screen_upscale (screen.h:126-149): replicate every pixel scale x scale into
   a surface of `pitch_bytes` per row (SDL_Surface->pitch / ->pixels).
   Host form (uploads sbuf... uses the last frame on the device if src is NULL)
   and device form. */
int pwn_screen_upscale(pwn_ctx *ctx, const uint32_t *sbuf, int scale, int pitch_bytes,
	uint32_t *pixels);
int pwn_upscale_device(pwn_ctx *ctx, const void *d_src, int scale, int pitch_bytes,
	void *d_dst, void *stream);

int pwn_get_stats(pwn_ctx *ctx, pwn_stats *out);

/* device-side probes of the arithmetic primitives (rcp/rsqrt tables, sinf,
   cosf, expf, sqrt, divide, colour pack, LCG); used by the parity tests.
   op: see PWN_PROBE_*; in/out are HOST arrays of n 32-bit words. */
#define PWN_PROBE_RCP    0
#define PWN_PROBE_RSQRT  1
#define PWN_PROBE_SINF   2
#define PWN_PROBE_COSF   3
#define PWN_PROBE_EXPF   4
#define PWN_PROBE_SQRT   5
#define PWN_PROBE_DIV    6  /* in = pairs (a,b), out = a/b */
#define PWN_PROBE_FTOINT 7  /* in = 4 floats per output word (util.h:48-59) */
#define PWN_PROBE_RANDFS 8  /* in = seed, out = randfs bits (util.h:13-16) */
#define PWN_PROBE_SIN_OF_PAIR 9   /* sinf / cosf halves of the joint sincos used for */
#define PWN_PROBE_COS_OF_PAIR 10  /* the floor normal (trace.h:45-46)                 */
int pwn_probe(pwn_ctx *ctx, int op, const uint32_t *in, uint32_t *out, int n);

#ifdef __cplusplus
}
#endif
#endif
