/* game_script.h -- the reference's shipped game.lua restated in C over the
 * object calls of include/pwnhip.h (Lua is not in this image); see game_script.c */
#ifndef PWN_GAME_SCRIPT_H
#define PWN_GAME_SCRIPT_H
#include "pwnhip.h"

#define GAME_MAX_OBJS 64

typedef struct game_script
{
	int n;
	double opos[GAME_MAX_OBJS][8];   /* dx dy dz r c1 c2 c3 refl  (game.lua:2-20) */
	int oball[GAME_MAX_OBJS];        /* object handles            (game.lua:1)    */
	double obx, oby, obz;            /* cluster centre            (game.lua:22)   */
	double obvx, obvz;               /* its heading               (game.lua:23)   */
} game_script;

/* script_newvm + the script's load-time body (script.h:71-103, game.lua:22-30):
   read the object table, create and place the objects.  0 or a PWN_* code / -100 (table unreadable) */
int game_script_init(game_script *g, pwn_ctx *ctx, const char *table_path);
/* on_tick(sec_current, sec_delta) (game.lua:33-87, called at main.c:127-140) */
int game_script_on_tick(game_script *g, pwn_ctx *ctx, double sec_current, double sec_delta);
#endif
