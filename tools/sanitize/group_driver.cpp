// The host logic of the row tiling and of the group (pwn_tiled.cpp, pwn_group.cpp, pwn_api.cpp) under ThreadSanitizer and
// AddressSanitizer + UBSan, on the CPU: fake HIP runtime, stand-in kernels (fake_kernels.cpp).  Every frame of a group of N
// members -- blocking calls, frames in flight delivered to the host, frames left on the devices -- must equal the frame of ONE
// context, bit for bit, with moving cuts, with a band of deep pixels that leaves the halo (repeat with whole strips), with
// depth that carries over, with a member that is late past the deadline (PWN_ETIMEDOUT, then recovery), and with
// processes over the shared-memory transport.  tools/sanitize/README.txt
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <sys/wait.h>
#include <vector>
#include "pwnhip.h"

static const char *LEVEL =
	"###########\r\n#;;;;;;;;;#\r\n#;;*;;;;;;#\r\n#;;;;$$;;;#\r\n#;a;;;;;b;#\r\n#;;;;;;;;;#\r\n###########\r\n";
#define CK(call) do { int rc_ = (call); if(rc_ < 0) { fprintf(stderr, "%s:%d %s -> %d (%s)\n", __FILE__, __LINE__, #call, rc_, ctx ? pwn_last_error(ctx) : ""); exit(1); } } while(0)

static void cam_for(float cam[16], int f)
{
	memset(cam, 0, 64);
	cam[0] = cam[5] = cam[10] = cam[15] = 1.0f;
	cam[12] = 3.5f + 0.01f * (float)f; cam[13] = 0.5f; cam[14] = 2.5f;
}

static void spheres_for(std::vector<pwn_sphere> &s, int f)
{
	s.clear();
	for(int i = 0; i < 5 + f % 3; i++) { pwn_sphere q = { 0.2f, 0.3f, 2.0f + 0.3f * (float)i + 0.01f * (float)f, 0.4f, 2.0f + 0.2f * (float)i, 0.1f, 0.5f, 0.9f }; s.push_back(q); }
}

// frame f's sec: from frame 20 on a band of deep pixels (fake_kernels.cpp) whose taps leave the halo
static float sec_for(int f) { return f < 20 ? 0.25f * (float)f : 100.0f + 0.7f * (float)f; }

static int run_frames(pwn_ctx *ctx, int w, int h, int frames, std::vector<std::vector<uint32_t>> &out, std::vector<std::vector<float>> &zout, int mode)
{
	const size_t n = (size_t)w * h;
	std::vector<pwn_sphere> sph;
	float cam[16];
	if(mode == 0)
	{
		std::vector<uint32_t> sb(n); std::vector<float> zb(n);
		for(int f = 0; f < frames; f++)
		{
			spheres_for(sph, f); cam_for(cam, f);
			CK(pwn_upload_spheres(ctx, sph.data(), (int)sph.size()));
			CK(pwn_trace_screen_centred(ctx, cam, sec_for(f), sb.data(), zb.data()));
			out.push_back(sb); zout.push_back(zb);
		}
		return 0;
	}
	// frames in flight: mode 1 delivered to the host, mode 2 left on the device(s), mode 3 delivered with the upscaled surface (pitch wider than the rows)
	const int scale = 2, pitch = w * scale * 4 + 16;
	if(mode == 3) CK(pwn_frames_config(ctx, 3, PWN_FRAME_SBUF | PWN_FRAME_SURFACE, scale, pitch));
	else CK(pwn_frames_config(ctx, 3, mode == 1 ? (PWN_FRAME_SBUF | PWN_FRAME_ZBUF) : 0, 1, 0));
	std::vector<uint32_t> tmp(n);
	for(int f = 0; f < frames + 3; f++)
	{
		if(f >= 3)
		{
			pwn_frame fr;
			CK(pwn_wait_frame(ctx, f % 3, &fr));
			if(mode == 1) { out.push_back(std::vector<uint32_t>(fr.sbuf, fr.sbuf + n)); zout.push_back(std::vector<float>(fr.zbuf, fr.zbuf + n)); }
			else if(mode == 3)
			{
				// (screen.h:132,138-139: a source row takes w*scale + pitch*(scale-1) words of the surface)
				std::vector<uint32_t> both(fr.sbuf, fr.sbuf + n);
				const size_t rowadv = (size_t)w * scale + (size_t)(pitch / 4) * (scale - 1);
				both.insert(both.end(), fr.surface, fr.surface + rowadv * (size_t)h);
				out.push_back(both);
			}
			else { CK(pwn_read_plane(ctx, fr.d_sbuf, tmp.data(), n * 4)); out.push_back(tmp); }
		}
		if(f < frames)
		{
			spheres_for(sph, f); cam_for(cam, f);
			CK(pwn_upload_spheres(ctx, sph.data(), (int)sph.size()));
			CK(pwn_submit_frame(ctx, cam, sec_for(f), f % 3));
		}
	}
	CK(pwn_frames_config(ctx, 0, 0, 1, 0));
	return 0;
}

static int same(const std::vector<std::vector<uint32_t>> &a, const std::vector<std::vector<uint32_t>> &b, const char *what, int members)
{
	if(a.size() != b.size()) { fprintf(stderr, "%s, %d members: %zu frames against %zu\n", what, members, a.size(), b.size()); return 1; }
	for(size_t f = 0; f < a.size(); f++)
		if(a[f] != b[f])
		{
			size_t i = 0;
			while(a[f][i] == b[f][i]) i++;
			fprintf(stderr, "%s, %d members: frame %zu differs first at pixel %zu\n", what, members, f, i);
			return 1;
		}
	return 0;
}

static int group_runs(int w, int h)
{
	const int frames = 30;
	pwn_ctx *ctx = NULL;
	std::vector<std::vector<uint32_t>> ref[4]; std::vector<std::vector<float>> zref[4];
	for(int mode = 0; mode < 4; mode++)
	{
		CK(pwn_init(&ctx, 0, w, h));
		CK(pwn_level_load_mem(ctx, LEVEL, (int)strlen(LEVEL)));
		run_frames(ctx, w, h, frames, ref[mode], zref[mode], mode);
		pwn_destroy(ctx); ctx = NULL;
	}
	int bad = 0;
	for(int members = 2; members <= 5; members++)
		for(int mode = 0; mode < 4; mode++)
		{
			int devs[8];
			for(int i = 0; i < members; i++) devs[i] = (mode == 2) ? i : 0;       // (distinct ordinals too: the peer-copy branch of the transport)
			CK(pwn_init_multi(&ctx, devs, members, w, h));
			CK(pwn_level_load_mem(ctx, LEVEL, (int)strlen(LEVEL)));
			std::vector<std::vector<uint32_t>> got; std::vector<std::vector<float>> zgot;
			run_frames(ctx, w, h, frames, got, zgot, mode);
			bad |= same(ref[mode], got, mode == 0 ? "blocking" : mode == 1 ? "delivered" : mode == 2 ? "resident" : "surface", members);
			// (depth: a pixel the trace leaves alone keeps the previous call's value, or the value of the frame that had the slot before)
			if(mode < 2) for(size_t f = 0; f < zgot.size(); f++) if(memcmp(zgot[f].data(), zref[mode][f].data(), zgot[f].size() * 4) != 0) { fprintf(stderr, "%s, %d members: depth of frame %zu differs\n", mode == 0 ? "blocking" : "delivered", members, f); bad = 1; break; }
			pwn_group_info gi;
			CK(pwn_group_info_get(ctx, &gi));
			// (a bounded halo is in force to begin with where the shortest strip of the equal split holds it: the deep band then forces a repeat)
			const int per = ((h + members - 1) / members + 7) / 8 * 8, last = h - per * (members - 1), halo = (int)(0.002 * h * 24.0) + 2;
			const bool bounded = (last < per ? last : per) >= halo;
			if(gi.members != members || gi.transport != PWN_TRANSPORT_LOCAL || (bounded && gi.frames_redone == 0)) { fprintf(stderr, "%d members: info members %d transport %d redone %llu\n", members, gi.members, gi.transport, (unsigned long long)gi.frames_redone); bad = 1; }
			pwn_stats st;
			CK(pwn_get_stats(ctx, &st));
			pwn_destroy(ctx); ctx = NULL;
		}
	printf("group runs %dx%d: %s\n", w, h, bad ? "FAILED" : "every frame of 2..5 members equals the one-context frame (blocking, delivered, resident)");
	return bad;
}

// a member that is late past the deadline: PWN_ETIMEDOUT (or the failure it causes in the others), then the handle works again
static int late_member(void)
{
	const int w = 64, h = 96;
	pwn_ctx *ctx = NULL;
	int devs[3] = { 0, 0, 0 };
	setenv("PWN_DBG_GROUP_STALL", "1:3:900", 1);
	CK(pwn_init_multi(&ctx, devs, 3, w, h));
	unsetenv("PWN_DBG_GROUP_STALL");
	CK(pwn_level_load_mem(ctx, LEVEL, (int)strlen(LEVEL)));
	CK(pwn_tiled_set_timeouts(ctx, 5000, 200));
	std::vector<uint32_t> sb((size_t)w * h), first;
	float cam[16];
	cam_for(cam, 0);
	int errors = 0;
	for(int f = 0; f < 6; f++)
	{
		const int rc = pwn_trace_screen_centred(ctx, cam, 0.5f, sb.data(), NULL);
		if(rc != PWN_OK) { errors++; printf("late member: call %d -> %d (%s)\n", f + 1, rc, pwn_last_error(ctx)); continue; }
		if(first.empty()) first = sb;
		else if(first != sb) { fprintf(stderr, "late member: frame %d differs\n", f); return 1; }
	}
	pwn_destroy(ctx);
	if(errors != 1) { fprintf(stderr, "late member: %d calls failed, expected exactly the third\n", errors); return 1; }
	printf("late member: one call failed at the deadline, the others delivered the same frame\n");
	return 0;
}

// processes over the shared-memory transport (pwn_tiled_* as a process per GPU uses it)
static int shm_ranks(int world)
{
	const int w = 64, h = 160, frames = 12;
	unsigned char id[PWN_TILED_ID_BYTES];
	pwn_ctx *ctx = NULL;
	CK(pwn_tiled_unique_id(id, PWN_TRANSPORT_SHM));
	std::vector<pid_t> kids;
	int rank = 0;
	fflush(NULL);
	for(int r = 1; r < world; r++) { pid_t k = fork(); if(k == 0) { rank = r; kids.clear(); break; } kids.push_back(k); }
	CK(pwn_init(&ctx, 0, w, h));
	CK(pwn_level_load_mem(ctx, LEVEL, (int)strlen(LEVEL)));
	CK(pwn_tiled_set_timeouts(ctx, 20000, 20000));
	CK(pwn_tiled_init(ctx, rank, world, id, PWN_TRANSPORT_SHM, -1));
	std::vector<pwn_sphere> sph;
	float cam[16];
	unsigned long long sum = 0;
	for(int f = 0; f < frames + 2; f++)
	{
		if(f < frames) { spheres_for(sph, f); cam_for(cam, f); CK(pwn_upload_spheres(ctx, sph.data(), (int)sph.size())); CK(pwn_tiled_submit(ctx, cam, sec_for(f + 14))); }
		if(f >= 2)
		{
			pwn_tiled_frame tf;
			CK(pwn_tiled_wait(ctx, PWN_TILED_HOST, &tf));
			if(rank == 0) for(size_t i = 0; i < (size_t)w * h; i += 13) sum += tf.sbuf[i];
		}
	}
	pwn_tiled_shutdown(ctx);
	pwn_destroy(ctx);
	if(rank != 0) _exit(0);
	int bad = 0;
	for(pid_t k : kids) { int st = 0; if(waitpid(k, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = 1; }
	printf("shm ranks %d: %s (checksum %llx)\n", world, bad ? "FAILED" : "done", sum);
	return bad;
}

int main(int argc, char **argv)
{
	setenv("PWN_GROUP_TRANSPORT", "local", 1);
	int bad = 0;
	const char *what = argc > 1 ? argv[1] : "all";
	if(!strcmp(what, "all") || !strcmp(what, "group")) { bad |= group_runs(64, 160); bad |= group_runs(128, 96); }
	if(!strcmp(what, "all") || !strcmp(what, "late")) bad |= late_member();
	if(!strcmp(what, "all") || !strcmp(what, "shm")) { bad |= shm_ranks(2); bad |= shm_ranks(3); }
	printf(bad ? "FAILED\n" : "ok\n");
	return bad;
}
