/*
 * player.c -- camera control and player motion of the reference's loop (main.c:142-379), restated
 * for hosts over the C ABI (see player.h; parity unpinned, stated there).
 * Compiled with -ffp-contract=off: every float product and sum below is rounded by itself.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "player.h"

#define PLAYER_BBOX 0.2f          /* defs.h:6 */

/* util.h:151-158 */
static int cell_at(const uint8_t *cells, int cx, int cz)
{
	if(cx < 0 || cx >= 64) cx = 0;
	if(cz < 0 || cz >= 64) cz = 0;
	return cells[cz * 64 + cx];
}

/* util.h:112-126: can a player at height y NOT stand in a cell of type c, coming from oldcell */
static int is_solid(const pwn_portal *pmap, int c, int oldcell, float y)
{
	if(c == '"' && (oldcell == '#' || oldcell == '&')) return y < 1.0f || y >= 2.0f;
	if(c == '#' || c == '&') return y < 0.0f || y >= 2.0f;
	if(c == ';' || c == '$' || c == '"') return y < 0.0f || y >= 1.0f;
	if(c == '>' || c == '<' || c == '^' || c == ',') return y < 0.0f || y >= 1.0f;
	if(c >= 'A' && c <= 'Z') return pmap[c - 'A'].x2 != -1 ? 0 : 1;
	return 1;
}

void pwn_player_init(pwn_player *p, const int32_t spawn[2])
{
	memset(p, 0, sizeof(*p));
	p->cam[0] = p->cam[5] = p->cam[10] = p->cam[15] = 1.0f;            /* mat4_iden */
	p->cam[12] = 0.5f + (float)spawn[0]; p->cam[13] = 0.5f; p->cam[14] = 0.5f + (float)spawn[1];
}

int pwn_key_from_name(const char *name)
{
	static const char *names[] = { "left", "right", "up", "down", "w", "s", "a", "d", "quit" };
	for(int i = 0; i < 9; i++) if(strcmp(name, names[i]) == 0) return i;
	return PWN_KEY_NONE;
}

void pwn_keys_event(pwn_keys *k, int sym, int down)
{
	down = down ? 1 : 0;
	switch(sym)
	{
		case PWN_KEY_LEFT: k->turnleft = down; break;
		case PWN_KEY_RIGHT: k->turnright = down; break;
		case PWN_KEY_UP: k->turnup = down; break;
		case PWN_KEY_DOWN: k->turndown = down; break;
		case PWN_KEY_W: k->moveforward = down; break;
		case PWN_KEY_S: k->moveback = down; break;
		case PWN_KEY_A: k->moveleft = down; break;
		case PWN_KEY_D: k->moveright = down; break;
		default: break;
	}
}

/* util.h:91-110 */
static void roty(float *cam, float ang)
{
	float vs = sinf(ang), vc = cosf(ang);
	float vxx = cam[0], vxz = cam[2], vzx = cam[8], vzz = cam[10];
	cam[0] = vc * vxx + vs * vxz;
	cam[2] = vc * vxz - vs * vxx;
	cam[8] = vc * vzx + vs * vzz;
	cam[10] = vc * vzz - vs * vzx;
}

void pwn_player_step(pwn_player *p, const pwn_keys *k, float tdiff, const uint8_t cells[4096], const pwn_portal pmap[26])
{
	float *cam = p->cam, *pos = p->cam + 12;
	roty(cam, tdiff * 3.0f * (float)(k->turnleft - k->turnright));          /* main.c:188 (up / down are read but unused there) */

	/* main.c:195-210: old cell, velocity along the forward and right rows, move */
	const int cx1 = (int)pos[0], cz1 = (int)pos[2];
	const float fwd = tdiff * 5.0f * (float)(k->moveforward - k->moveback);
	const float side = tdiff * 5.0f * (float)(k->moveleft - k->moveright);
	float vel[4];
	for(int i = 0; i < 4; i++) vel[i] = cam[8 + i] * fwd + cam[i] * side;
	for(int i = 0; i < 4; i++) pos[i] += vel[i];

	/* main.c:212-266: push back out of solid cells, by the corner of the bounding box that leads */
	const float px1 = pos[0], py1 = pos[1], pz1 = pos[2];
	const int gx1 = vel[0] < 0.0f ? -1 : 1, gz1 = vel[2] < 0.0f ? -1 : 1;
	const int bcx = (int)(px1 + (float)gx1 * PLAYER_BBOX), bcz = (int)(pz1 + (float)gz1 * PLAYER_BBOX);
	const int oldcell = cell_at(cells, cx1, cz1);
	const float backx = (float)cx1 + 0.5f + (0.5f - PLAYER_BBOX) * (float)gx1;
	const float backz = (float)cz1 + 0.5f + (0.5f - PLAYER_BBOX) * (float)gz1;
	if(cx1 != bcx && cz1 != bcz)
	{
		const int solx = is_solid(pmap, cell_at(cells, bcx, cz1), oldcell, py1);
		const int solz = is_solid(pmap, cell_at(cells, cx1, bcz), oldcell, py1);
		const int solc = is_solid(pmap, cell_at(cells, bcx, bcz), oldcell, py1);
		if(solx && solz) { pos[0] = backx; pos[2] = backz; }
		else if(solx) pos[0] = backx;
		else if(solz) pos[2] = backz;
		else if(solc) pos[2] = backz;
	}
	else if(cx1 != bcx)
	{
		if(is_solid(pmap, cell_at(cells, bcx, bcz), oldcell, py1)) pos[0] = backx;
	}
	else if(cz1 != bcz)
	{
		if(is_solid(pmap, cell_at(cells, bcx, bcz), oldcell, py1)) pos[2] = backz;
	}

	/* main.c:268-276: gravity */
	for(int i = 0; i < 4; i++) pos[i] += p->gravity[i];
	p->gravity[1] -= 3.0f * tdiff * tdiff;
	if(pos[1] < 0.4f) { pos[1] = 0.4f; p->gravity[1] = 0.0f; }

	/* main.c:278-378: a new cell: steps of a two-high room, portals */
	const int cx2 = (int)pos[0], cz2 = (int)pos[2];
	if(cx1 == cx2 && cz1 == cz2) return;
	const int c1 = cell_at(cells, cx1, cz1), c2 = cell_at(cells, cx2, cz2);
	if((c1 == '#' || c1 == '&') && c2 == '"') pos[1] -= 1.0f;
	else if(c1 == '"' && (c2 == '#' || c2 == '&')) pos[1] += 1.0f;
	else if(c2 >= 'A' && c2 <= 'Z')
	{
		const pwn_portal *pm = &pmap[c2 - 'A'];
		p->traversals++;
		int rot = 0;
		float rx = pos[0], rz = pos[2], rcx = (float)cx2, rcz = (float)cz2;
		float rvxx = cam[0], rvxz = cam[2], rvzx = cam[8], rvzz = cam[10];
		if(pm->x2 == -1) { /* an unpaired letter: nothing happens (main.c:308-310) */ }
		else if(pm->x1 == cx2 && pm->z1 == cz2)
		{
			rx += (float)((cx2 - cx1) + (pm->x2 - pm->x1));
			rz += (float)((cz2 - cz1) + (pm->z2 - pm->z1));
			rcx = (float)pm->x2; rcz = (float)pm->z2;
			rot = (-pm->rot12) & 3;
		}
		else if(pm->x2 == cx2 && pm->z2 == cz2)
		{
			rx += (float)((cx2 - cx1) - (pm->x2 - pm->x1));
			rz += (float)((cz2 - cz1) - (pm->z2 - pm->z1));
			rcx = (float)pm->x1; rcz = (float)pm->z1;
			rot = pm->rot12 & 3;
		}
		const float trx = rx, trz = rz, trvxx = rvxx, trvxz = rvxz, trvzx = rvzx, trvzz = rvzz;
		switch(rot)
		{
			case 1:
				rx = (rcx + 0.5f) + (trz - (rcz + 0.5f));
				rz = (rcz + 0.5f) - (trx - (rcx + 0.5f));
				rvxx = trvxz; rvxz = -trvxx; rvzx = trvzz; rvzz = -trvzx;
				break;
			case 2:
				rx = (rcx + 0.5f) * 2.0f - rx;
				rz = (rcz + 0.5f) * 2.0f - rz;
				rvxx = -trvxx; rvxz = -trvxz; rvzx = -trvzx; rvzz = -trvzz;
				break;
			case 3:
				rx = (rcx + 0.5f) - (trz - (rcz + 0.5f));
				rz = (rcz + 0.5f) + (trx - (rcx + 0.5f));
				rvxx = -trvxz; rvxz = trvxx; rvzx = -trvzz; rvzz = trvzx;
				break;
			default: break;
		}
		pos[0] = rx; pos[2] = rz;
		cam[0] = rvxx; cam[2] = rvxz; cam[8] = rvzx; cam[10] = rvzz;
	}
}

int pwn_keys_load(const char *path, pwn_key_event *ev, int cap)
{
	FILE *fp = fopen(path, "r");
	if(fp == NULL) return -1;
	int n = 0;
	char line[256];
	while(fgets(line, sizeof(line), fp) != NULL)
	{
		char *hash = strchr(line, '#');
		if(hash != NULL) *hash = 0;
		int frame;
		char key[32], state[32];
		if(line[strspn(line, " \t\r\n")] == 0) continue;               /* blank or comment */
		int got = sscanf(line, "%d %31s %31s", &frame, key, state);
		int sym = got == 3 ? pwn_key_from_name(key) : PWN_KEY_NONE;
		if(got != 3 || sym == PWN_KEY_NONE || (strcmp(state, "down") != 0 && strcmp(state, "up") != 0) || frame < 0) { fclose(fp); return -1; }
		if(n < cap) { ev[n].frame = frame; ev[n].sym = sym; ev[n].down = strcmp(state, "down") == 0; }
		n++;
	}
	fclose(fp);
	return n;
}
