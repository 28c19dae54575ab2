// pwn_internal.h -- the context behind include/pwnhip.h, shared by the translation
// units of libpwnhip.so (pwn_api.cpp: frames, tables; pwn_tiled.cpp: row tiling over RCCL).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include <atomic>

#include "pwnhip.h"
#include "tables.h"

struct pwn_blur_params
{
	int w, h, y0, y1;
	int groups;
	const uint32_t *pre;
	const float *zbuf;
	uint32_t *out;
	const uint2 *skip;
	int avail_y0, avail_y1;
	uint32_t *miss;
	int tile_h, tile_w, batch;
	uint32_t *cost_acc, *cost_out;
	uint32_t cost_mul, cost_div;   // ... scaled on the way: the resident grid over the grid the trace ran with (PWN_OPT_TRACE_ROOM), so that ranks with and without room compare
};

extern "C" hipError_t pwn_launch_trace(const pwn_trace_params *P, int grid, size_t lds_bytes, bool count, hipStream_t stream);
extern "C" hipError_t pwn_launch_trace_refill(const pwn_trace_params *P, int grid, size_t lds_bytes, bool count, hipStream_t stream);
extern "C" unsigned pwn_trace_refill_lds_extra(bool has_w);
extern "C" int pwn_trace_refill_blocks_per_cu(size_t lds_bytes, bool count, bool has_w);
extern "C" int pwn_trace_blocks_per_cu(size_t lds_bytes, bool count, bool has_w);
extern "C" int pwn_trace_tile_h(void);
extern "C" int pwn_trace_tile_w(void);
extern "C" unsigned pwn_trace_lds_extra(void);
extern "C" hipError_t pwn_launch_blur(const pwn_blur_params *P, hipStream_t stream);
extern "C" hipError_t pwn_launch_order(const uint16_t *cost, uint32_t units, uint32_t cap, uint32_t *perm, hipStream_t stream);
extern "C" hipError_t pwn_launch_upscale(const uint32_t *src, uint32_t *dst, int w, int h, int scale, int pitch, hipStream_t stream);
extern "C" hipError_t pwn_launch_upload(const void *h_pinned_src, void *d_dst, size_t bytes, hipStream_t stream);
extern "C" hipError_t pwn_launch_words(const uint32_t *d_all, const uint32_t *d_own, uint32_t *h_pinned_dst, int world, hipStream_t stream);
extern "C" hipError_t pwn_launch_probe(int op, const uint32_t *in, uint32_t *out, int n, const uint16_t *tabs, hipStream_t stream);

// LDS budget for the table blob: leave room so that at least two workgroups
// fit per CU (160 KiB LDS per CU on gfx950)
#define PWN_BLOB_MAX (72u * 1024u)
#ifndef PWN_NBLOB
#define PWN_NBLOB 4       // device copies of the blob: an upload never touches one that launches in flight read.  With two, the upload
                          // of frame f+1's tables had to wait for the end of frame f-1 and so ran right where trace f starts;
                          // with three or four it runs at once, somewhere beside the trace grid: 0.3823 -> 0.3790 ms per 4K frame
#endif
#define PWN_NCOUNTERS 48     // device counters of the counting kernel variants: 16 (pwn_stats) + 24 regions + spare
#define PWN_TILED_STREAMS_DEFAULT 2
#define PWN_TICKET_SETS 6u  // launch n counts in set n mod 2R and clears set (n + R) mod 2R, R = pwn_ctx.launch_rot streams in rotation, 2 or 3 (pwn_i_launch_trace)
#define PWN_NSTAGE 4      // pinned staging buffers for those uploads
#define PWN_CALL_STRIPS_MAX 32   // row strips of one blocking call at most
#define PWN_HOST_REGS_MAX 16     // host buffers a context keeps registered (pwn_host_register)

// one frame in flight (pwn_submit_frame / pwn_wait_frame)
struct pwn_slot
{
	uint32_t *d_out; float *d_z; uint32_t *d_surface;      // device: final colour, depth, upscaled surface
	uint32_t *h_sbuf; float *h_zbuf; uint32_t *h_surface;  // pinned host copies handed to the caller
	hipEvent_t ev_k[4];                                    // compute stream: start, after trace, after blur, after sink
	hipEvent_t ev_done;                                    // copy stream: this frame's host buffers are complete
	bool in_flight, timed, beside;                         // (beside: launched while frames alternate between two streams)
	float sec;
	uint64_t seq;
};

// PWN_OPT_TRACE_ROOM: how many workgroups the persistent trace grid leaves free while frames alternate between two compute
// streams, so that the OTHER stream's kernels (the previous frame's blur, the next frame's grid) find room on every CU instead
// of waiting for this grid's end.  Worth +3.5 % at 4K and -11 % kernel time on the strips of an 8-way tiling on level.txt,
// and -3 % on synth256: so the default measures -- pwn_room_frame_done() is told of every delivered frame,
// times a window of frames with no room and one with a workgroup per CU, keeps the better for a while, and looks again.
struct pwn_room_ctl
{
	int mode;                 // -1: measure (default); >= 0: that many workgroups, always
	int arm, best;            // 0 = no room, 1 = one workgroup per CU
	int skip, hold;           // delivered frames not counted after a switch; frames left before the next look
	double sum[2]; int cnt[2];
	double t_prev, t_hold;
	unsigned long long looks, switches;
};
int pwn_room_for_launch(struct pwn_ctx *c);          // workgroups to leave free for a launch on one of two alternating streams
void pwn_room_frame_done(struct pwn_ctx *c);         // a frame of such a sequence was delivered to the host

struct pwn_tiled;        // pwn_tiled.cpp
struct pwn_group;        // pwn_group.cpp: several contexts of one process behind one handle (pwn_init_multi)

// What the members of a group share besides their handle (pwn_group.cpp makes it, pwn_tiled.cpp uses it): a meeting point for the
// members' threads with a deadline, a flag that ends every wait when a member has failed, the members themselves (every member
// reads the others' two words of a frame straight from their pinned memory: one process, no exchange needed for eight bytes),
// and the mailboxes of the in-process transport (PWN_TRANSPORT_LOCAL).
#define PWN_HUB_RING 64
// One box per ordered pair (from -> to).  A message is a rendezvous: the receiver posts where the bytes are to go and an event
// behind whatever of its own still uses that place; the sender, on ITS stream, waits for that event, copies, and says so with an
// event the receiver's stream then waits for.  So a send is complete, in the sender's stream order, when the bytes have left --
// the sender may overwrite its buffer behind it, as with ncclSend.
struct pwn_hub_post { void *dst; size_t bytes; hipEvent_t ev_free; int device; };
struct pwn_hub_box
{
	std::atomic<unsigned long long> posted, copied, consumed;      // by the receiver; by the sender (<= posted); by the receiver (<= copied)
	pwn_hub_post post[PWN_HUB_RING];
	hipEvent_t done[PWN_HUB_RING];
};
struct pwn_hub
{
	int world;
	std::atomic<int> failed;                     // a member gave up (an error, a deadline): nobody waits for it any more
	std::atomic<unsigned> bar_count, bar_gen;
	struct pwn_ctx *member[PWN_TILED_MAX_WORLD];
	pwn_hub_box *boxes;                          // world x world, [from * world + to]
};
int pwn_hub_meet(pwn_hub *h, int wait_ms);       // every member's thread arrives, or PWN_ETIMEDOUT / PWN_EHIP (failed)

struct pwn_ctx
{
	int device, w, h;
	int num_cus;
	int blur_passes, counters_on, scheduler, refill_limit;
	bool have_level;

	uint8_t cells[4096];
	uint32_t cell_base[65 * 65];      // the level's part of every cell word (char + class bits; row / column 64 = the clamp copies): built
	bool cell_base_ok;               // by pack_blob when the level changed, patched per upload with the cells' sphere lists
	pwn_portal pmap[26];
	int32_t spawn[2];
	std::vector<pwn_sphere> spheres;          // the live spheres the current lists index
	std::vector<int32_t> bin_off, bin_idx;
	std::vector<pwn_sphere> objs;             // lv->objs (defs.h:98): every slot ever handed out
	std::vector<uint8_t> obj_typ;             // P_INVAL / P_FREE / P_SPHERE (defs.h:55-60)

	std::vector<uint8_t> blob;       // host image of the LDS blob
	// The device copy exists twice: an upload goes into the buffer the launches in flight do not
	// read, on its own stream and from a pinned staging ring, so the host never waits for the GPU
	// between frames (level_prepare_render runs every frame, main.c:95).
	uint8_t *d_blob[PWN_NBLOB]; int blob_cur;
	bool blob_has_static[PWN_NBLOB];                                   // the rcp / rsqrt tables are in place
	hipEvent_t ev_tables[PWN_NBLOB]; bool tables_in_use[PWN_NBLOB];    // behind the last trace launch reading that copy
	hipEvent_t tables_wait[PWN_NBLOB];                                 // ... the event to wait for: ev_tables[i], or the caller's (below)
	// Set by a caller of pwn_i_launch_trace for its next launch: an event the caller records itself right behind
	// that launch anyway (its frame's "kernels done").  The launcher then records none of its own -- every
	// event between two kernels costs the queue a few microseconds.  The caller must have waited on the host for
	// the frame of the event's previous record (a slot is handed in again only when it is free): the launcher
	// drops the guards that still point at the event before the caller records it again.
	hipEvent_t trace_tables_event;
	hipEvent_t ev_upload[PWN_NBLOB]; bool upload_pending[PWN_NBLOB];   // behind the last upload into that copy
	uint8_t *h_stage[PWN_NSTAGE]; hipEvent_t ev_stage[PWN_NSTAGE]; bool stage_used[PWN_NSTAGE]; unsigned stage_next;
	hipStream_t up_stream;
	uint16_t tabs[4096];             // expanded rcp + rsqrt tables (never change)
	uint32_t off_sph;
	uint32_t off_recsph; int dbg_sphere_lists;      // inline sphere records in the per-cell lists (tables.h): where their "which sphere" array is, 0 = indexed lists; PWN_SPHERE_LISTS
	bool blob_dirty;
	size_t occ_lds[8]; int occ_blocks[8];    // cached occupancy query per kernel variant

	uint32_t *d_pre, *d_out;         // pre-blur ("tsbuf") and final ("sbuf") frames
	float *d_z;
	uint2 *d_skip;                   // blur LCG skip-ahead, w/4 entries
	unsigned long long *d_counters;
	unsigned long long *d_wave_log; int wave_log_on; size_t wave_log_cap;   // PWN_OPT_WAVE_LOG; entries (16 B) allocated
	// PWN_OPT_UNIT_ORDER: per stream that trace launches go out on ([0] `stream`, [1] `stream2`, [2..3] a strip-form caller's own)
	// what every unit of the last trace launch cost its wave (pwn_trace_params.unit_cost, indexed by the unit's number in
	// arithmetic order) and, sorted from that behind the frame's blur (pwn_i_launch_order), the order the NEXT launch of the same
	// rows on that stream hands its units out in.  Everything that touches one entry is ordered by ITS stream: the launch that
	// writes the costs, the sort that reads them and writes the order, the next launch that reads the order.
	struct unit_order_state
	{
		hipStream_t key; bool used; unsigned long long stamp;                 // the stream this entry belongs to
		uint16_t *d_cost; uint32_t *d_perm; size_t cost_cap, perm_cap;        // allocated: u16 entries, u32 entries
		uint32_t units, qcap; int y0, y1;                                     // what d_cost holds: of a launch over rows [y0, y1)
		bool cost_fresh;                                                      // written by a launch, not sorted yet
		uint32_t perm_units; int perm_y0, perm_y1; bool perm_valid;           // what d_perm orders
	} order[4];
	unsigned long long order_stamp;
	int tiled_choreo;                // PWN_OPT_TILED_CHOREO, read by pwn_tiled_init
	int tiled_streams;               // PWN_OPT_TILED_STREAMS, read by pwn_tiled_init
	int tiled_comms;                 // PWN_OPT_TILED_COMMS, read by pwn_tiled_init
	int unit_order;                  // the option: 1 = units handed out by last launch's cost, 0 = arithmetic order
	unsigned long long order_used, order_sorts;      // trace launches that ran in a sorted order; sorts launched (pwn_unit_order_state)
	bool dbg_force_hasw; int dbg_blocks_per_cu;      // PWN_DBG_* hooks, read at pwn_init
	int dbg_blur_th, dbg_blur_tw, dbg_blur_batch;                 // PWN_DBG_BLUR_TH: tile height of every blur launch (8 / 16 / 32), 0 = the launcher's choice
	int grid_reserve;                // workgroups the persistent trace grid leaves free (row tiling over RCCL), else 0
	uint32_t *trace_cost_word;       // likewise: pwn_trace_params.cost_word for the next launch
	uint32_t *trace_clear_word;      // set by a caller of pwn_i_launch_trace for its next launch: see pwn_trace_params.clear_word
	uint32_t *d_tickets; unsigned ticket_set;  // PWN_TICKET_SETS sets of work-queue counters of the trace kernel, used in turn
	uint32_t *d_scratch; size_t scratch_cap;   // upscale / probe staging

	hipStream_t stream;              // compute
	hipStream_t stream2;             // frames in flight alternate between `stream` and this one (PWN_OPT_FRAME_OVERLAP)
	uint32_t *d_pre2;                // the pre-blur plane of the frames on stream2 (allocated with the first of them)
	uint32_t cost_mul, cost_div;     // set by pwn_i_launch_trace: resident grid / grid it launched (what its cost word is to be scaled by); read by the tiling
	uint32_t blur_cost_mul, blur_cost_div;   // ... handed to the next pwn_i_launch_blur by its caller (0 = 1 / 1)
	pwn_room_ctl room; int launch_room;   // PWN_OPT_TRACE_ROOM; what the next trace launch leaves free (set by its caller, cleared by the launch)
	int frame_overlap;               // PWN_OPT_FRAME_OVERLAP
	hipEvent_t last_frame_done; hipStream_t last_frame_stream;   // "kernels done" of the frame submitted last, and its stream
	hipStream_t copy_stream;         // frames in flight: D2H of finished frames
	hipEvent_t ev[4];
	// The blocking call in row strips (PWN_OPT_CALL_STRIPS; pwn_api.cpp, call_in_strips): trace strip k, blur strip k - 1 from the
	// rows traced so far, strip k - 2 on its way to the caller's buffer -- the copy over PCIe is twice a 4K frame's kernels
	int call_strips;                 // the option: -1 = by frame size (default), 0 = one launch per pass, n >= 2 = that many strips
	hipStream_t copy_stream2;        // the blocking call in strips: chunks go out on one copy stream or on the two in turn
	int strip_copy_streams;          // ... 1 or 2 as found out by the context's first sixteen calls in strips (0: still finding out); PWN_CALL_COPY_STREAMS
	int strip_calib_n; double strip_calib_ms[16];
	hipEvent_t strip_ev[2 * PWN_CALL_STRIPS_MAX]; int strip_ev_n;      // behind strip k's trace [2k] and blur [2k + 1]; made on first use
	uint32_t *d_strip_miss, *h_strip_miss;     // taps of a strip's blur that left the rows traced so far: the device word, its pinned copy
	int strips_last;                 // strips of the last blocking call (1 = one launch per pass)
	int strip_reach;                 // 0: chunks are blurred where taps of depth <= 8 are traced, 1: <= 24 (after a frame whose taps went further)
	int strip_backoff;               // blocking calls left before strips are tried again after such a frame
	unsigned long long strip_calls, strip_redone;      // blocking calls that ran in strips; ... whose blur was repeated over the whole frame
	struct host_reg { void *base; size_t bytes; } host_regs[PWN_HOST_REGS_MAX]; int host_regs_n;     // pwn_host_register
	pwn_stats stats;

	// frames in flight
	int nslots, frame_flags, frame_scale, frame_pitch, frame_timing;
	pwn_slot slot[PWN_MAX_SLOTS];
	uint64_t frame_seq;

	pwn_group *grp; bool grp_head;   // pwn_init_multi: the group this context is the handle of (grp_head) or a member of; NULL otherwise
	pwn_hub *hub;                    // ... and what its members share (NULL outside a group)
	pwn_tiled *tiled;                // row tiling over RCCL, NULL until pwn_tiled_init
	int tiled_init_ms, tiled_wait_ms;   // pwn_tiled_set_timeouts (0 = the default: environment, else 120 s / 60 s)

	// The trace launch before the last one: the stream it went on and an event behind it.  Launch n clears the ticket set
	// of launch n + 2 = the set launch n - 2 drew from, so it has to come after launch n - 2: true by itself on one
	// stream and while frames alternate between two (n - 2 is then on n's stream), NOT when a launch leaves that pattern
	// (a blocking call or a counted frame between alternating ones) -- pwn_i_launch_trace then waits for this event.
	hipStream_t launch_stream[3]; hipEvent_t launch_event[3];     // [0] the last launch, [1] the one before, [2] the one before that
	unsigned long long launch_waits; // launches that left the rotation and were put behind the launch R before them (pwn_launch_order_waits)
	int launch_rot;                  // streams that successive trace launches rotate over: 2 (one stream, or two alternating), 3 (pwn_i_set_launch_rotation)

	char err[256];
};

#define HIPCHK(ctx, call) do { hipError_t e_ = (call); if(e_ != hipSuccess) { \
	snprintf((ctx)->err, sizeof((ctx)->err), "%s: %s", #call, hipGetErrorString(e_)); return PWN_EHIP; } } while(0)

// pwn_api.cpp internals used by pwn_tiled.cpp
int pwn_i_launch_trace(pwn_ctx *c, const float cam[16], float sec, int y0, int y1, uint32_t *d_sbuf, float *d_zbuf, hipStream_t stream);
int pwn_i_launch_blur(pwn_ctx *c, int y0, int y1, const uint32_t *d_pre, const float *d_z, uint32_t *d_out, hipStream_t stream,
	int avail_y0, int avail_y1, uint32_t *d_miss, uint32_t *d_cost_acc, uint32_t *d_cost_out);
void pwn_launch_history_clear(pwn_ctx *c);      // the launch-order events are about to be destroyed (streams idle)
int pwn_i_set_launch_rotation(pwn_ctx *c, int rot);
int pwn_i_launch_order(pwn_ctx *c, hipStream_t stream);      // behind a frame's last kernel on `stream`: sort that stream's unit costs (no-op when there are none)
void pwn_tiled_destroy(pwn_ctx *c);
bool pwn_tiled_busy(pwn_ctx *c);        // frames of the row tiling in flight
// pwn_tiled.cpp internals used by pwn_group.cpp: a member's frame with the host buffers its strip is delivered into (NULL: the
// frame stays on the devices and is gathered on member 0), and the switch that makes the tiling deliver to the host
int pwn_i_tiled_submit(pwn_ctx *c, const float cam[16], float sec, uint32_t *host_sbuf, float *host_zbuf, int zplane, uint32_t *host_surface);
int pwn_i_tiled_surface(pwn_ctx *c, int scale, int pitch_bytes);
int pwn_i_tiled_sink(pwn_ctx *c);
int pwn_i_tiled_ready(pwn_ctx *c, int ahead);
// pwn_api.cpp: level_prepare_render's binning of a compact list of live spheres (no context: PWN_ETOOBIG where the lists would not
// fit on chip), and its upload into one context; the object table of pwn_upload_spheres by itself
struct pwn_binned { std::vector<pwn_sphere> s; std::vector<int32_t> off, idx; };
int pwn_i_bin_spheres(const pwn_sphere *s, int n, pwn_binned *out);
int pwn_i_upload_binned(pwn_ctx *c, const pwn_binned &b);
void pwn_i_set_object_table(pwn_ctx *c, const pwn_sphere *s, int n);
// pwn_group.cpp: the entry points of include/pwnhip.h when the context is a group's handle
int pwn_group_set_option(pwn_ctx *h, int option, int value);
int pwn_group_level_mem(pwn_ctx *h, const char *text, int len);
int pwn_group_upload_level(pwn_ctx *h, const uint8_t *data, const pwn_portal *pmap);
int pwn_group_upload_spheres(pwn_ctx *h, const pwn_sphere *s, int n);
int pwn_group_prepare_render(pwn_ctx *h);
int pwn_group_trace_screen_centred(pwn_ctx *h, const float cam[16], float sec, uint32_t *sbuf, float *zbuf);
int pwn_group_frames_config(pwn_ctx *h, int nslots, int flags, int scale, int pitch_bytes);
int pwn_group_submit_frame(pwn_ctx *h, const float cam[16], float sec, int slot);
int pwn_group_wait_frame(pwn_ctx *h, int slot, pwn_frame *out);
int pwn_group_frame_ready(pwn_ctx *h, int slot);
int pwn_group_get_stats(pwn_ctx *h, pwn_stats *out);
int pwn_group_screen_upscale(pwn_ctx *h, const uint32_t *sbuf, int scale, int pitch_bytes, uint32_t *pixels);
int pwn_group_host_register(pwn_ctx *h, void *base, size_t bytes);
int pwn_group_host_unregister(pwn_ctx *h, void *base);
int pwn_group_set_timeouts(pwn_ctx *h, int init_ms, int wait_ms);
pwn_ctx *pwn_group_member(pwn_ctx *h, int i);                 // member i (0 = the one that holds the level and the object table)
int pwn_group_sync(pwn_ctx *h);                               // everything posted to the members is through
#include <functional>
int pwn_group_on_member0(pwn_ctx *h, const std::function<int(pwn_ctx *)> &fn);      // ... on member 0's thread
void pwn_group_destroy(pwn_ctx *h);
