/* player.h -- the interactive half of the reference's frame loop for hosts over the C ABI:
   key state (main.c:142-186) and the camera's motion per tick (main.c:188-379: turning, walking
   with push-back from solid cells, gravity, stepping between the levels of a two-high room,
   walking through a portal).  Plain C, no GPU: it works on the level tables pwn_get_level hands out.
   PARITY UNPINNED: main.c cannot be built in this image (SDL 1.2, Lua 5.1) and the reference has
   no fixtures for it; tests/test_player.py checks it against the pinned ray path instead (a walk
   through a portal shortens the view ray by the distance walked) and against its own invariants. */
#ifndef PWN_PLAYER_H
#define PWN_PLAYER_H
#include <stdint.h>
#include "pwnhip.h"

/* the eight key flags of main.c:72-79 */
typedef struct pwn_keys { int turnleft, turnright, turnup, turndown, moveforward, moveback, moveleft, moveright; } pwn_keys;

typedef struct pwn_player
{
	float cam[16];          /* mat4 rows x, y, z, w = right, up, forward, position (defs.h:46-52) */
	float gravity[4];       /* main.c:48,59 */
	int traversals;         /* portals walked through so far (main.c:296 prints each) */
} pwn_player;

/* an SDL key event as the loop reads it (main.c:142-186): sym is one of PWN_KEY_*, down 1 / 0 */
enum { PWN_KEY_LEFT, PWN_KEY_RIGHT, PWN_KEY_UP, PWN_KEY_DOWN, PWN_KEY_W, PWN_KEY_S, PWN_KEY_A, PWN_KEY_D, PWN_KEY_QUIT, PWN_KEY_NONE };

void pwn_player_init(pwn_player *p, const int32_t spawn[2]);                 /* main.c:59-64 */
int pwn_key_from_name(const char *name);                                    /* "left" "right" "up" "down" "w" "s" "a" "d" "quit" */
void pwn_keys_event(pwn_keys *k, int sym, int down);                        /* main.c:147-185 */
/* one pass of main.c:188-379 for a tick of tdiff seconds */
void pwn_player_step(pwn_player *p, const pwn_keys *k, float tdiff, const uint8_t cells[4096], const pwn_portal pmap[26]);

/* A key script: text lines `FRAME KEY down|up` (# comments), FRAME = the frame after whose render the
   event is polled (main.c:142).  Returns the number of events read into ev (at most cap), -1 on a bad line or file. */
typedef struct pwn_key_event { int frame, sym, down; } pwn_key_event;
int pwn_keys_load(const char *path, pwn_key_event *ev, int cap);
#endif
