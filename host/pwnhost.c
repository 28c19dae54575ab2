/*
 * pwnhost.c -- a plain-C host over libpwnhip.so, shaped like the reference's
 * frame loop (main.c:93-109): prepare objects, trace_screen_centred,
 * screen_upscale, present.  It is the drop-in demonstration of
 * include/pwnhip.h: no HIP, no C++ and no Python on this side of the ABI.
 *
 * SDL and Lua are not in this image, so "present" writes the upscaled
 * surface as a binary PPM (the role of SDL_Flip, main.c:109).  The objects
 * come either from a text file of obj_set "sphere" arguments in obj_set's
 * own order (script.h:10-40), one sphere per line,
 *     r refl x y z  b g r
 * or (-g) from the shipped game script restated in C (game_script.c), which
 * is ticked once per frame like main.c:127-140 does.
 *
 *   pwnhost level.txt [-s spheres.txt | -g game_objects.txt] [-w W] [-h H]
 *           [-x SCALE] [-n FRAMES] [-t SECONDS_PER_FRAME] [-a TURN_PER_FRAME]
 *           [-p PITCH] [-b BLUR_PASSES] [-o out.ppm] [-d DEVICE] [-v 1] [-q SLOTS]
 *
 * -k keys.txt makes it the interactive loop without a window: the camera is the player's (main.c:188-379:
 * turning, walking with push-back, gravity, portals), driven by key events from a script
 * (`FRAME KEY down|up`, keys left right up down w s a d quit; polled after frame FRAME like
 * SDL_PollEvent at main.c:142).  -a is ignored then.
 * -S 1 (hosts built with SDL 1.2: host/Makefile finds sdl-config) opens the reference's window
 * (main.c:386-394), shows every frame (SDL_Flip, main.c:109) and takes the keys from it (main.c:142-186:
 * arrows turn, w s a d walk, closing the window quits): the playable loop; the clock is the wall clock
 * like main.c:112-114 unless -t is given.  The blocking loop only (no -q / -W).
 * -l game.lua (hosts built with Lua 5.1: host/Makefile) runs a script by path in a Lua VM whose
 * obj_new / obj_set / obj_free / level_get act on the library's object table (script.h:1-103).
 * -t fixes the clock step (the reference uses wall time, main.c:112-114), which
 * makes a run reproducible; -v 1 prints every frame's hash.
 * -W WORLD alone row-tiles every frame over WORLD processes that this one forks (rank r on device r);
 * -W WORLD -R RANK -I IDFILE [-T rccl|shm] row-tiles every frame over WORLD processes, one per
 * GPU (pwn_tiled_*: RCCL inside the library): start the same command once per rank with its
 * -R (and -d DEVICE); rank 0 writes the group id to IDFILE, the others wait for it; rank 0
 * presents the frames.  -H ROWS sets the halo (default -1 = depth 24, 0 = whole strips).
 * -M 1: the frames are delivered to the host by every rank (pwn_tiled_host_sink): one frame buffer in POSIX
 * shared memory, every rank copies its strip into it over its own PCIe link, nothing is gathered to rank 0.
 * -O 1: the gather's root rotates over the ranks (pwn_tiled_gather_root): frame f is assembled, and presented, on rank f mod WORLD.
 * -G N: the frame row-tiled over N GPUs by THIS process (pwn_init_multi): one handle, the loops below unchanged -- the blocking
 * loop of main.c:93-109, or with -q the frames in flight; devices 0..N-1, or N members on the one device of -d (boxes with one GPU).
 * -X SECONDS / -Y SECONDS: how long the tiling's bring-up / a wait for a frame may wait for the other ranks (pwn_tiled_set_timeouts;
 * defaults 120 / 60); a rank that does not answer ends the run with the library's message and exit status 2, not with a hang.
 * -q SLOTS (2..4) keeps that many frames in flight (pwn_submit_frame /
 * pwn_wait_frame): the frame and its upscaled surface arrive in the library's
 * pinned host buffers while the next frame's kernels run; frame f is presented
 * SLOTS-1 frames late, the pixels are the same.
 *
 * Build: make -C host      (gcc only; links libpwnhip.so by path)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include <time.h>
#include <fcntl.h>
#include <unistd.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>

#include "pwnhip.h"
#include "game_script.h"
#include "player.h"
#ifdef HAVE_LUA
#include "lua_host.h"
#endif
#ifdef HAVE_SDL
#include <SDL.h>
#endif

/* the globals a reference host owns (main.c:26-34) */
static int rwidth = 320, rheight = 200, rscale = 3;
static uint32_t *sbuf = NULL;
static float *zbuf = NULL;
static float sec_current = 0.0f;
static struct { int pitch; uint32_t *pixels; } surface;

static double now_s(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* util.h:61-71 / 91-110: camera = identity turned about y */
static void cam_identity(float cam[16])
{
	memset(cam, 0, 16 * sizeof(float));
	cam[0] = cam[5] = cam[10] = cam[15] = 1.0f;
}

static void cam_roty(float cam[16], float ang)
{
	float vs = sinf(ang), vc = cosf(ang);
	float vxx = cam[0], vxz = cam[2], vzx = cam[8], vzz = cam[10];
	cam[0] = vc * vxx + vs * vxz;
	cam[2] = vc * vxz - vs * vxx;
	cam[8] = vc * vzx + vs * vzz;
	cam[10] = vc * vzz - vs * vzx;
}

/* the camera of a frame: the player's (-k), or mainloop's start pose turned about y (-a) */
static void frame_camera(float cam[16], const pwn_player *player, float ang, const int32_t spawn[2])
{
	if(player != NULL) { memcpy(cam, player->cam, 16 * sizeof(float)); return; }
	cam_identity(cam);
	if(ang != 0.0f) cam_roty(cam, ang);
	cam[12] = 0.5f + (float)spawn[0]; cam[13] = 0.5f; cam[14] = 0.5f + (float)spawn[1];
}

/* main.c:142-379 for the frame just shown: the key events polled now, then the player's tick */
static int player_tick(pwn_player *player, pwn_keys *keys, const pwn_key_event *ev, int nev, int frame, float tdiff,
	const uint8_t *cells, const pwn_portal *pmap)
{
	for(int i = 0; i < nev; i++)
		if(ev[i].frame == frame)
		{
			if(ev[i].sym == PWN_KEY_QUIT) return 1;
			pwn_keys_event(keys, ev[i].sym, ev[i].down);
		}
	pwn_player_step(player, keys, tdiff, cells, pmap);
	return 0;
}

static int load_spheres(const char *path, pwn_sphere **out)
{
	FILE *fp = fopen(path, "r");
	if(fp == NULL) return -1;
	int n = 0, cap = 16;
	pwn_sphere *s = malloc(sizeof(*s) * (size_t)cap);
	char line[512];
	while(s != NULL && fgets(line, sizeof(line), fp) != NULL)
	{
		pwn_sphere t;
		if(line[0] == '#') continue;
		if(sscanf(line, "%f %f %f %f %f %f %f %f", &t.r, &t.refl, &t.x, &t.y, &t.z, &t.cb, &t.cg, &t.cr) != 8) continue;
		if(n == cap) { cap *= 2; s = realloc(s, sizeof(*s) * (size_t)cap); if(s == NULL) break; }
		s[n++] = t;
	}
	fclose(fp);
	*out = s;
	return s == NULL ? -1 : n;
}

static int write_ppm(const char *path, const uint32_t *px, int w, int h, int pitch_words)
{
	FILE *fp = fopen(path, "wb");
	if(fp == NULL) return -1;
	fprintf(fp, "P6\n%d %d\n255\n", w, h);
	unsigned char *row = malloc((size_t)w * 3);
	for(int y = 0; y < h && row != NULL; y++)
	{
		for(int x = 0; x < w; x++)
		{
			uint32_t p = px[(size_t)y * (size_t)pitch_words + (size_t)x];   /* BGRA8 (util.h:48-59) */
			row[3 * x + 0] = (unsigned char)(p >> 16);
			row[3 * x + 1] = (unsigned char)(p >> 8);
			row[3 * x + 2] = (unsigned char)p;
		}
		fwrite(row, 3, (size_t)w, fp);
	}
	free(row);
	fclose(fp);
	return 0;
}

static uint64_t fnv64(const uint32_t *p, size_t n)
{
	uint64_t h = 1469598103934665603ULL;
	for(size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ULL; }
	return h;
}

/* print-and-return, the reference's error model (level.h:35-37,110-115).  A deadline of the row tiling (a rank that does not
   answer, pwn_tiled_set_timeouts) leaves the process at once: a helper thread of the bring-up may still sit inside RCCL, and
   tearing the runtime down under it is not worth the risk */
#define CHK(call) do { int rc_ = (call); if(rc_ != PWN_OK) { \
	fprintf(stderr, "%s -> %s (%d): %s\n", #call, pwn_strerror(rc_), rc_, pwn_last_error(ctx)); \
	if(rc_ == PWN_ETIMEDOUT) { fflush(NULL); _exit(2); } \
	pwn_destroy(ctx); return 1; } } while(0)

int main(int argc, char **argv)
{
	const char *level = NULL, *sphfile = NULL, *gamefile = NULL, *out = NULL, *luafile = NULL, *keyfile = NULL;
	int frames = 1, device = 0, blur = 1, pitch = 0, verbose = 0, slots = 0;
	int world = 1, rank = 0, halo = -1, transport = PWN_TRANSPORT_RCCL, window = 0, frames_given = 0, hostsink = 0;
	float tiled_init_s = 0.0f, tiled_wait_s = 0.0f;
	int rotate_root = 0, group = 0;
	int rank_given = 0, device_given = 0;
	const char *idfile = NULL, *nonce = "";
	const time_t started = time(NULL);
	float turn = 0.0f, fixed_dt = -1.0f;
	for(int i = 1; i < argc; i++)
	{
		if(argv[i][0] != '-') { level = argv[i]; continue; }
		if(i + 1 >= argc) { fprintf(stderr, "option %s needs a value\n", argv[i]); return 2; }
		switch(argv[i][1])
		{
			case 's': sphfile = argv[++i]; break;
			case 'g': gamefile = argv[++i]; break;
			case 'l': luafile = argv[++i]; break;
			case 'k': keyfile = argv[++i]; break;
			case 'S': window = atoi(argv[++i]); break;
			case 't': fixed_dt = (float)atof(argv[++i]); break;
			case 'v': verbose = atoi(argv[++i]); break;
			case 'w': rwidth = atoi(argv[++i]); break;
			case 'h': rheight = atoi(argv[++i]); break;
			case 'x': rscale = atoi(argv[++i]); break;
			case 'n': frames = atoi(argv[++i]); frames_given = 1; break;
			case 'a': turn = (float)atof(argv[++i]); break;
			case 'p': pitch = atoi(argv[++i]); break;
			case 'b': blur = atoi(argv[++i]); break;
			case 'o': out = argv[++i]; break;
			case 'd': device = atoi(argv[++i]); device_given = 1; break;
			case 'q': slots = atoi(argv[++i]); break;
			case 'W': world = atoi(argv[++i]); break;
			case 'R': rank = atoi(argv[++i]); rank_given = 1; break;
			case 'I': idfile = argv[++i]; break;
			case 'N': nonce = argv[++i]; break;
			case 'H': halo = atoi(argv[++i]); break;
			case 'M': hostsink = atoi(argv[++i]); break;
			case 'O': rotate_root = atoi(argv[++i]); break;
			case 'G': group = atoi(argv[++i]); break;
			case 'T': transport = strcmp(argv[++i], "shm") == 0 ? PWN_TRANSPORT_SHM : PWN_TRANSPORT_RCCL; break;
			case 'X': tiled_init_s = (float)atof(argv[++i]); break;
			case 'Y': tiled_wait_s = (float)atof(argv[++i]); break;
			default: fprintf(stderr, "unknown option %s\n", argv[i]); return 2;
		}
	}
	if(level == NULL)
	{
		fprintf(stderr, "usage: pwnhost level.txt [-s spheres.txt | -g game_objects.txt] [-w W] [-h H] [-x SCALE] "
			"[-n FRAMES] [-t SEC_PER_FRAME] [-a TURN] [-p PITCH_BYTES] [-b BLUR] [-o out.ppm] [-d DEVICE] [-v 1] [-q SLOTS]\n"
			"       [-W WORLD -R RANK -I IDFILE [-N LAUNCH_NONCE] [-T rccl|shm] [-H HALO_ROWS] [-M 1 | -O 1]]\n"
			"       [-G GPUS_OF_THIS_PROCESS]\n");
		return 2;
	}
	/* -W WORLD without -R: this process starts the other ranks itself (fork, before anything touches a GPU): rank r
	   on device r (-d DEVICE: all on that one, for boxes with one GPU and -T shm), the id in a file of its own */
	pid_t kids[64];
	int nkids = 0;
	char idbuf[64];
	if(world > 1 && !rank_given)
	{
		if(world > 64) { fprintf(stderr, "-W takes at most 64 ranks\n"); return 2; }
		if(idfile == NULL) { snprintf(idbuf, sizeof(idbuf), "/tmp/pwnhost_%d.id", (int)getpid()); idfile = idbuf; unlink(idfile); }
		for(int r = 1; r < world && rank == 0; r++)
		{
			pid_t k = fork();
			if(k < 0) { perror("fork"); return 1; }
			if(k == 0) { rank = r; nkids = 0; }
			else kids[nkids++] = k;
		}
		if(!device_given) device = rank;
	}
	if(world < 1 || rank < 0 || rank >= world || (world > 1 && idfile == NULL))
	{
		fprintf(stderr, "-W WORLD: either alone (the ranks are forked), or with -R RANK (0..WORLD-1) and -I IDFILE once per rank\n");
		return 2;
	}
	if(slots != 0 && (slots < 2 || slots > PWN_MAX_SLOTS)) { fprintf(stderr, "-q takes 2..%d\n", PWN_MAX_SLOTS); return 2; }
#ifndef HAVE_SDL
	if(window) { fprintf(stderr, "-S: this pwnhost was built without SDL 1.2 (host/Makefile)\n"); return 2; }
#endif
	if(window && (slots != 0 || world > 1)) { fprintf(stderr, "-S goes with the blocking loop only\n"); return 2; }
	if(window && !frames_given) frames = 0x7fffffff;          /* a window runs until it is closed (SDL_QUIT, main.c:145) */
#ifndef HAVE_LUA
	if(luafile != NULL) { fprintf(stderr, "-l %s: this pwnhost was built without Lua 5.1 (host/Makefile); use -g\n", luafile); return 2; }
#endif
	if(slots != 0 && fixed_dt < 0.0f) fixed_dt = 0.0f;       /* frames in flight run on a fixed clock step (-t) */
	if(rscale < 1) rscale = 1;
	if(pitch == 0) pitch = rwidth * rscale * 4;

	/* main.c:395-400 */
	size_t npix = (size_t)rwidth * (size_t)rheight;
	sbuf = malloc(npix * 4);
	zbuf = malloc(npix * 4);
	surface.pitch = pitch;
	surface.pixels = calloc((size_t)pitch * (size_t)rheight * (size_t)rscale, 1);
	if(sbuf == NULL || zbuf == NULL || surface.pixels == NULL) { fprintf(stderr, "out of memory\n"); return 1; }

	pwn_ctx *ctx = NULL;
	int rc;
	if(group > 1)
	{
		/* one process, one handle, GROUP devices behind it (main.c's one loop stays what it is) */
		int devs[PWN_TILED_MAX_WORLD];
		if(group > PWN_TILED_MAX_WORLD || world > 1) { fprintf(stderr, "-G takes 2..%d GPUs of this one process (not with -W)\n", PWN_TILED_MAX_WORLD); return 2; }
		for(int i = 0; i < group; i++) devs[i] = device_given ? device : i;
		rc = pwn_init_multi(&ctx, devs, group, rwidth, rheight);
		if(rc != PWN_OK) { fprintf(stderr, "pwn_init_multi: %s (%d)\n", pwn_strerror(rc), rc); return 1; }
		if(tiled_init_s > 0.0f || tiled_wait_s > 0.0f) CHK(pwn_tiled_set_timeouts(ctx, (int)(tiled_init_s * 1000.0f), (int)(tiled_wait_s * 1000.0f)));
		/* (main.c:395-400's buffers, made known to the devices once: every device copies its strip straight into them) */
		CHK(pwn_host_register(ctx, sbuf, npix * 4));
		CHK(pwn_host_register(ctx, zbuf, npix * 4));
	}
	else rc = pwn_init(&ctx, device, rwidth, rheight);
	if(rc != PWN_OK) { fprintf(stderr, "pwn_init: %s (%d)\n", pwn_strerror(rc), rc); return 1; }
	CHK(pwn_set_option(ctx, PWN_OPT_BLUR_PASSES, blur));

	/* main.c:51-64 */
	CHK(pwn_level_load(ctx, level));
	int32_t spawn[2];
	CHK(pwn_get_level(ctx, NULL, NULL, spawn));
	printf("spawn: %d %d\n", spawn[0], spawn[1]);
	pwn_player player;
	pwn_keys keys;
	static pwn_key_event key_ev[4096];
	static uint8_t cells[4096];
	static pwn_portal pmap[26];
	int nkey_ev = 0;
	memset(&keys, 0, sizeof(keys));
	pwn_player_init(&player, spawn);
	const int interactive = keyfile != NULL || window;
	if(interactive) CHK(pwn_get_level(ctx, cells, pmap, NULL));
	if(keyfile != NULL)
	{
		nkey_ev = pwn_keys_load(keyfile, key_ev, 4096);
		if(nkey_ev < 0 || nkey_ev > 4096) { fprintf(stderr, "cannot read the key script %s\n", keyfile); pwn_destroy(ctx); return 1; }
		if(fixed_dt < 0.0f) fixed_dt = 1.0f / 60.0f;          /* a scripted run is reproducible: a fixed clock step */
	}
#define PLAYER(frame, dt) (interactive && player_tick(&player, &keys, key_ev, nkey_ev, (frame), (dt), cells, pmap))
	pwn_sphere *sph = NULL;
	int nsph = 0;
	if(sphfile != NULL && (nsph = load_spheres(sphfile, &sph)) < 0) { fprintf(stderr, "cannot read %s\n", sphfile); pwn_destroy(ctx); return 1; }
#ifdef HAVE_LUA
	lua_host *vm = NULL;
	char luaerr[512];
	if(luafile != NULL)                                                          /* script_newvm, main.c:56 */
	{
		vm = lua_host_new(ctx, luafile, luaerr, sizeof(luaerr));
		if(vm == NULL) { fprintf(stderr, "ERROR: %s\n", luaerr); pwn_destroy(ctx); return 1; }
	}
#define TICK(sec, dt) do { if(vm != NULL && lua_host_on_tick(vm, (sec), (dt), luaerr, sizeof(luaerr)) != 0) { \
	fprintf(stderr, "ERROR: %s\n", luaerr); pwn_destroy(ctx); return 1; } } while(0)
#else
#define TICK(sec, dt) do { } while(0)
#endif
	game_script game;
	if(luafile != NULL) { /* the script created its objects */ }
	else if(gamefile != NULL)                                                    /* script_newvm, main.c:56 */
	{
		rc = game_script_init(&game, ctx, gamefile);
		if(rc == -100) { fprintf(stderr, "cannot read %s\n", gamefile); pwn_destroy(ctx); return 1; }
		CHK(rc);
	}
	else
		CHK(pwn_upload_spheres(ctx, sph, nsph));

	float ang = 0.0f;
	double t_first = 0.0, t_rest = 0.0;
	if(world > 1)
	{
		/* the frame loop of main.c:93-140 with every frame row-tiled over `world` processes; two frames in
		   flight (three), the present step two frames behind; every rank runs the same loop (and the same script) */
		/* The group id travels through IDFILE, and an id belongs to ONE launch: a file left behind by an earlier run of
		   the same command would hand ranks 1.. the previous run's id, and rank 0 would wait in ncclCommInitRank for
		   peers that never come.  So the file starts with a line "pwnid <nonce>": the launcher gives every rank of a
		   launch the same -N NONCE (anything unique: its pid, a timestamp) and the other ranks take only a file that
		   carries it; without -N they take only a file written after they started themselves, less a minute for
		   launchers that start the ranks one by one.  Rank 0 removes IDFILE before it writes the new one and again
		   when it is done. */
		unsigned char id[PWN_TILED_ID_BYTES];
		char head[128];
		snprintf(head, sizeof(head), "pwnid %.100s\n", nonce);
		const size_t hlen = strlen(head);
		if(rank == 0)
		{
			unlink(idfile);
			CHK(pwn_tiled_unique_id(id, transport));
			char tmp[1024];
			snprintf(tmp, sizeof(tmp), "%s.tmp", idfile);
			FILE *fp = fopen(tmp, "wb");
			if(fp == NULL || fwrite(head, 1, hlen, fp) != hlen || fwrite(id, 1, sizeof(id), fp) != sizeof(id)) { fprintf(stderr, "cannot write %s\n", tmp); pwn_destroy(ctx); return 1; }
			fclose(fp);
			rename(tmp, idfile);
		}
		else
		{
			int got = 0;
			for(int tries = 0; tries < 12000 && !got; tries++)
			{
				struct stat sb;
				FILE *fp = fopen(idfile, "rb");
				char have[128];
				if(fp != NULL && fstat(fileno(fp), &sb) == 0 && (nonce[0] != 0 || sb.st_mtime >= started - 60) &&
				   fread(have, 1, hlen, fp) == hlen && memcmp(have, head, hlen) == 0 && fread(id, 1, sizeof(id), fp) == sizeof(id)) got = 1;
				if(fp != NULL) fclose(fp);
				if(!got) { struct timespec ts = { 0, 10 * 1000 * 1000 }; nanosleep(&ts, NULL); }
			}
			if(!got) { fprintf(stderr, "rank %d: no group id of this launch in %s (a stale file of an earlier run? see -N)\n", rank, idfile); pwn_destroy(ctx); return 1; }
		}
		if(tiled_init_s > 0.0f || tiled_wait_s > 0.0f) CHK(pwn_tiled_set_timeouts(ctx, (int)(tiled_init_s * 1000.0f), (int)(tiled_wait_s * 1000.0f)));
		CHK(pwn_tiled_init(ctx, rank, world, id, transport, halo));
		void *host_frames = NULL;
		const size_t host_bytes = (size_t)PWN_TILED_SLOTS * npix * 4;
		char shm_name[256];
		shm_name[0] = 0;
		if(hostsink)
		{
			/* one frame buffer for all ranks: POSIX shared memory named after the id file.  Rank 0 creates it;
			   the others, who leave pwn_tiled_init (a collective) after rank 0 entered it, wait for its size. */
			const char *b = strrchr(idfile, '/');
			snprintf(shm_name, sizeof(shm_name), "/pwn_frames_%s", b ? b + 1 : idfile);
			int fd = -1;
			if(rank == 0)
			{
				fd = shm_open(shm_name, O_CREAT | O_RDWR, 0600);
				if(fd >= 0 && ftruncate(fd, (off_t)host_bytes) != 0) { close(fd); fd = -1; }
			}
			else
				for(int tries = 0; tries < 12000; tries++)
				{
					struct stat sb;
					fd = shm_open(shm_name, O_RDWR, 0600);
					if(fd >= 0 && fstat(fd, &sb) == 0 && (size_t)sb.st_size >= host_bytes) break;
					if(fd >= 0) { close(fd); fd = -1; }
					struct timespec ts = { 0, 10 * 1000 * 1000 };
					nanosleep(&ts, NULL);
				}
			if(fd < 0) { fprintf(stderr, "rank %d: shared memory %s\n", rank, shm_name); pwn_destroy(ctx); return 1; }
			host_frames = mmap(NULL, host_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
			close(fd);
			if(host_frames == MAP_FAILED) { fprintf(stderr, "rank %d: mmap %s\n", rank, shm_name); pwn_destroy(ctx); return 1; }
			CHK(pwn_tiled_host_sink(ctx, host_frames, host_bytes));
		}
		if(rotate_root && !hostsink) CHK(pwn_tiled_gather_root(ctx, PWN_TILED_ROOT_ROTATE));
		if(fixed_dt < 0.0f) fixed_dt = 0.0f;
		pwn_tiled_frame tf;
		memset(&tf, 0, sizeof(tf));
		double t0 = now_s(), t1 = t0;
		for(int f = 0; f < frames + 2; f++)
		{
			if(f < frames)
			{
				float cam[16];
				frame_camera(cam, interactive ? &player : NULL, ang, spawn);
				CHK(pwn_prepare_render(ctx));                                           /* main.c:95 */
				CHK(pwn_tiled_submit(ctx, cam, sec_current));                          /* main.c:107 */
			}
			if(f >= 2)
			{
				CHK(pwn_tiled_wait(ctx, PWN_TILED_HOST, &tf));
				if(tf.sbuf != NULL && verbose)                 /* (the frame's root -- rank 0 unless -O 1 --, or every rank with a host sink) */
					printf("frame %d sec %.9g fnv64 %016llx\n", f - 2, (double)(fixed_dt * (float)(f - 2)), (unsigned long long)fnv64(tf.sbuf, npix));
				if(f == 2) t1 = now_s();
			}
			if(f < frames)
			{
				sec_current += fixed_dt;                                                /* main.c:112-114 */
				if(gamefile != NULL)
					CHK(game_script_on_tick(&game, ctx, (double)sec_current, (double)fixed_dt));   /* main.c:127-140 */
				TICK((double)sec_current, (double)fixed_dt);
				if(PLAYER(f, fixed_dt)) frames = f + 1;                                /* SDL_QUIT, main.c:145 */
				ang += turn;
			}
		}
		double t2 = now_s();
		pwn_tiled_info inf;
		CHK(pwn_tiled_get_info(ctx, &inf));
		printf("rank %d of %d: rows [%d,%d) now (the cuts moved %llu times; %d rows to begin with), halo %d rows, %llu frames (%llu repeated with whole strips), %llu grouped exchanges, sent %.1f MB, received %.1f MB\n",
			inf.rank, inf.world, inf.y0, inf.y1, (unsigned long long)inf.recuts, inf.rows_per_rank, inf.halo_rows, (unsigned long long)inf.frames,
			(unsigned long long)inf.frames_redone, (unsigned long long)inf.groups, (double)inf.bytes_sent / 1e6, (double)inf.bytes_received / 1e6);
		if(hostsink ? rank == 0 : tf.sbuf != NULL)         /* whoever holds the last frame presents it */
		{
			CHK(pwn_screen_upscale(ctx, tf.sbuf, rscale, surface.pitch, surface.pixels));      /* main.c:108 */
			printf("frame %dx%d x%d: sbuf fnv64 %016llx, surface fnv64 %016llx\n", rwidth, rheight, rscale,
				(unsigned long long)fnv64(tf.sbuf, npix),
				(unsigned long long)fnv64(surface.pixels, (size_t)(pitch / 4) * (size_t)rheight * (size_t)rscale));
			if(frames > 1)
				printf("tiled loop: %.2f Mpixels/s over %d frames (first frame %.1f ms)\n",
					(double)npix * (frames - 1) / (t2 - t1) / 1e6, frames - 1, (t1 - t0) * 1e3);
			if(out != NULL && write_ppm(out, surface.pixels, rwidth * rscale, rheight * rscale, pitch / 4) != 0)
				fprintf(stderr, "cannot write %s\n", out);
		}
		pwn_tiled_shutdown(ctx);
		pwn_destroy(ctx);
		if(host_frames != NULL)
		{
			munmap(host_frames, host_bytes);
			if(rank == 0) shm_unlink(shm_name);
		}
		free(sph); free(sbuf); free(zbuf); free(surface.pixels);
		/* the ranks this process started */
		int bad = 0;
		for(int i = 0; i < nkids; i++)
		{
			int st = 0;
			if(waitpid(kids[i], &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = 1;
		}
		if(rank == 0) unlink(idfile);                /* (every rank has read it: pwn_tiled_init is a collective) */
		return bad;
	}
	if(slots > 0)
	{
		/* frames in flight: the loop of main.c:93-140 with the present step SLOTS-1 frames behind */
		CHK(pwn_frames_config(ctx, slots, PWN_FRAME_SBUF | PWN_FRAME_SURFACE, rscale, pitch));
		pwn_frame fr;
		memset(&fr, 0, sizeof(fr));
		double t0 = now_s(), t1 = t0;
		for(int f = 0; f < frames + slots - 1; f++)
		{
			if(f >= slots - 1)
			{
				int k = f - (slots - 1);
				CHK(pwn_wait_frame(ctx, k % slots, &fr));                              /* SDL_Flip's place, main.c:109 */
				if(verbose)
					printf("frame %d sec %.9g fnv64 %016llx\n", k, (double)fr.sec_current, (unsigned long long)fnv64(fr.sbuf, npix));
				if(k == 0) t1 = now_s();
			}
			if(f < frames)
			{
				float cam[16];
				frame_camera(cam, interactive ? &player : NULL, ang, spawn);
				CHK(pwn_prepare_render(ctx));                                           /* main.c:95 */
				CHK(pwn_submit_frame(ctx, cam, sec_current, f % slots));                /* main.c:107-108 */
				sec_current += fixed_dt;                                                /* main.c:112-114 */
				if(gamefile != NULL)
					CHK(game_script_on_tick(&game, ctx, (double)sec_current, (double)fixed_dt));   /* main.c:127-140 */
				TICK((double)sec_current, (double)fixed_dt);
				if(PLAYER(f, fixed_dt)) frames = f + 1;                                /* SDL_QUIT, main.c:145 */
				ang += turn;
			}
		}
		double t2 = now_s();
		printf("frame %dx%d x%d: sbuf fnv64 %016llx, surface fnv64 %016llx\n", rwidth, rheight, rscale,
			(unsigned long long)fnv64(fr.sbuf, npix),
			(unsigned long long)fnv64(fr.surface, (size_t)(pitch / 4) * (size_t)rheight * (size_t)rscale));
		printf("last frame on the device: trace %.3f ms, blur %.3f ms, sink %.3f ms\n", fr.trace_ms, fr.blur_ms, fr.sink_ms);
		if(frames > 1)
			printf("host loop incl. upload, D2H and sink, %d frames in flight: %.2f Mpixels/s over %d frames (first frame %.1f ms)\n",
				slots, (double)npix * (frames - 1) / (t2 - t1) / 1e6, frames - 1, (t1 - t0) * 1e3);
		if(out != NULL && write_ppm(out, fr.surface, rwidth * rscale, rheight * rscale, pitch / 4) != 0)
			fprintf(stderr, "cannot write %s\n", out);
		pwn_destroy(ctx);
		free(sph); free(sbuf); free(zbuf); free(surface.pixels);
		return 0;
	}
#ifdef HAVE_SDL
	SDL_Surface *screen = NULL;
	if(window)                                                                  /* main.c:386-394 */
	{
		if(SDL_Init(SDL_INIT_VIDEO | SDL_INIT_TIMER) != 0 || (screen = SDL_SetVideoMode(rwidth * rscale, rheight * rscale, 32, 0)) == NULL)
		{
			fprintf(stderr, "SDL: %s\n", SDL_GetError());
			pwn_destroy(ctx);
			return 1;
		}
		SDL_WM_SetCaption("pwnfps on libpwnhip", NULL);
	}
#endif
	for(int f = 0; f < frames; f++)
	{
		float cam[16];
		frame_camera(cam, interactive ? &player : NULL, ang, spawn);

		double t0 = now_s();
		CHK(pwn_prepare_render(ctx));                                           /* level_prepare_render, main.c:95 */
		CHK(pwn_trace_screen_centred(ctx, cam, sec_current, sbuf, zbuf));       /* main.c:107 */
		CHK(pwn_screen_upscale(ctx, NULL, rscale, surface.pitch, surface.pixels)); /* main.c:108 */
		double dt = now_s() - t0;
		if(f == 0) t_first = dt; else t_rest += dt;

		if(verbose)
			printf("frame %d sec %.9g fnv64 %016llx\n", f, (double)sec_current, (unsigned long long)fnv64(sbuf, npix));
		float tdiff = fixed_dt >= 0.0f ? fixed_dt : (float)dt;
		sec_current += tdiff;                                                    /* main.c:112-114 */
		if(gamefile != NULL)
			CHK(game_script_on_tick(&game, ctx, (double)sec_current, (double)tdiff));   /* main.c:127-140 */
		TICK((double)sec_current, (double)tdiff);
#ifdef HAVE_SDL
		if(window)
		{
			/* SDL_Flip, main.c:109: the sink's surface into the window's, row by row (the pitches may differ) */
			if(SDL_MUSTLOCK(screen)) SDL_LockSurface(screen);
			for(int y = 0; y < rheight * rscale; y++)
				memcpy((unsigned char *)screen->pixels + (size_t)y * screen->pitch, (unsigned char *)surface.pixels + (size_t)y * (size_t)surface.pitch,
					(size_t)rwidth * (size_t)rscale * 4);
			if(SDL_MUSTLOCK(screen)) SDL_UnlockSurface(screen);
			SDL_Flip(screen);
			SDL_Event ev;                                                          /* main.c:142-186 */
			while(SDL_PollEvent(&ev))
			{
				if(ev.type == SDL_QUIT) frames = f + 1;
				if(ev.type != SDL_KEYDOWN && ev.type != SDL_KEYUP) continue;
				int sym = PWN_KEY_NONE;
				switch(ev.key.keysym.sym)
				{
					case SDLK_LEFT: sym = PWN_KEY_LEFT; break;
					case SDLK_RIGHT: sym = PWN_KEY_RIGHT; break;
					case SDLK_UP: sym = PWN_KEY_UP; break;
					case SDLK_DOWN: sym = PWN_KEY_DOWN; break;
					case SDLK_w: sym = PWN_KEY_W; break;
					case SDLK_s: sym = PWN_KEY_S; break;
					case SDLK_a: sym = PWN_KEY_A; break;
					case SDLK_d: sym = PWN_KEY_D; break;
					default: break;
				}
				pwn_keys_event(&keys, sym, ev.type == SDL_KEYDOWN);
			}
		}
#endif
		if(PLAYER(f, tdiff)) frames = f + 1;                                        /* SDL_QUIT, main.c:145 */
		ang += turn;
	}
#ifdef HAVE_SDL
	if(window) SDL_Quit();
#endif

	pwn_stats st;
	CHK(pwn_get_stats(ctx, &st));
	if(group > 1)
	{
		pwn_group_info gi;
		CHK(pwn_group_info_get(ctx, &gi));
		printf("group of %d in one process, exchange over %s: rows", gi.members, gi.transport == PWN_TRANSPORT_RCCL ? "RCCL" : "copies inside the process");
		for(int i = 0; i <= gi.members; i++) printf(" %d", gi.cuts[i]);
		printf(" (the cuts moved %llu times), halo %d rows, %llu frames (%llu repeated with whole strips)\n", (unsigned long long)gi.recuts, gi.halo_rows,
			(unsigned long long)gi.frames, (unsigned long long)gi.frames_redone);
	}
	printf("frame %dx%d x%d: sbuf fnv64 %016llx, surface fnv64 %016llx\n", rwidth, rheight, rscale,
		(unsigned long long)fnv64(sbuf, npix),
		(unsigned long long)fnv64(surface.pixels, (size_t)(pitch / 4) * (size_t)rheight * (size_t)rscale));
	printf("last frame on the device: trace %.3f ms, blur %.3f ms, with D2H %.3f ms\n", st.trace_ms, st.blur_ms, st.total_ms);
	if(frames > 1)
		printf("host loop incl. upload, D2H and sink: %.2f Mpixels/s over %d frames (first frame %.1f ms)\n",
			(double)npix * (frames - 1) / t_rest / 1e6, frames - 1, t_first * 1e3);
	if(out != NULL && write_ppm(out, surface.pixels, rwidth * rscale, rheight * rscale, pitch / 4) != 0)
		fprintf(stderr, "cannot write %s\n", out);

	pwn_destroy(ctx);
	free(sph); free(sbuf); free(zbuf); free(surface.pixels);
	return 0;
}
