/* CPU sanitizer run of host/player.c (see README.txt): random levels, random key sequences and tick lengths
   (including absurd ones), thousands of ticks each; the camera must stay finite or the run says where not. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include "pwnhip.h"
#include "level_host.h"
#include "player.h"
int main(void)
{
	srand(4242);
	static const char alpha[] = ";;;;;;$$##&&\"\"<>,^...**\n\n\nABCABCDDEE ";
	long nonfinite = 0, trav = 0;
	for(int it = 0; it < 600; it++)
	{
		int len = 200 + rand() % 5000;
		char *t = malloc(len + 1);
		for(int i = 0; i < len; i++) t[i] = alpha[rand() % (sizeof(alpha) - 1)];
		/* a spawn point somewhere */
		t[rand() % len] = '@';
		uint8_t cells[4096]; pwn_portal pm[26]; int32_t sp[2];
		if(pwn_parse_level(t, len, cells, pm, sp) != 0) return 1;
		free(t);
		pwn_player p; pwn_keys k; memset(&k, 0, sizeof(k));
		pwn_player_init(&p, sp);
		for(int tick = 0; tick < 3000; tick++)
		{
			if(rand() % 7 == 0) pwn_keys_event(&k, rand() % 10, rand() & 1);
			float dt = (float)(rand() % 1000) / 20000.0f;
			if(rand() % 500 == 0) dt = 3.0f;             /* a stall of the host */
			if(rand() % 5000 == 0) dt = 1e6f;
			pwn_player_step(&p, &k, dt, cells, pm);
			int ok = 1;
			for(int i = 0; i < 16; i++) if(!isfinite(p.cam[i])) ok = 0;
			if(!ok) { nonfinite++; pwn_player_init(&p, sp); }
		}
		trav += p.traversals;
	}
	printf("fuzz_player: 600 levels x 3000 ticks, %ld portal traversals, %ld resets after a non-finite camera\n", trav, nonfinite);
	return 0;
}
