/*
 * lv2_plugin.c -- the LV2 front end of the MI355X phase rotator: the drop-in
 * for src/phaserotate.c's plugin interface.  Same exported symbol
 * (lv2_descriptor), same URIs, port indices, required features, atom
 * vocabulary, latency and error behaviour; every DSP operation goes through
 * the C ABI of libprot_hip.so (include/prot_hip.h).  Host-side work kept here,
 * as in the reference: port plumbing, the atom control/notify protocol and the
 * peak meters (src/phaserotate.c:451-509, :573-611, :727-771, :800-830).
 *
 * There is no CPU DSP in this file: if the engine cannot be created
 * (no GPU), instantiate() fails exactly like any other instantiate() failure
 * of the reference: message on stderr, NULL.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lv2_abi.h"
#include "prot_hip.h"

#define PHASEROTATE_URI "http://gareus.org/oss/lv2/phaserotate" /* src/phaserotate.h:21 */
#define PROT_URI PHASEROTATE_URI "#"
#define MAX_CHANNELS 2 /* src/phaserotate.h:97 */

/* src/phaserotate.h:99-111 -- port indices are ABI */
enum {
	PORT_ATOM_CONTROL = 0,
	PORT_ATOM_NOTIFY  = 1,
	PORT_LATENCY      = 2,
	PORT_ANGLE0       = 3,
	PORT_INPUT0       = 4,
	PORT_OUTPUT0      = 5
};

/* URID vocabulary, src/phaserotate.h:41-93 */
typedef struct {
	LV2_URID atom_Blank, atom_Object, atom_Float, atom_Int, atom_Bool, atom_Sequence;
	LV2_URID ui_on, ui_off, reset_peaks, state, s_uiscale, s_link, levels;
	LV2_URID l_channel, l_in_cur, l_in_mom, l_in_peak, l_out_cur, l_out_mom, l_out_peak;
	LV2_URID l_diff_cur, l_diff_min, l_diff_max;
} Urids;

typedef struct {
	float momentary, peak;
	int   hold;
} Meter;

typedef struct {
	const float* in;
	float*       out;
	const float* angle;
	float*       dly;        /* last `latency` input frames, for the input meter */
	Meter        mtr[2];     /* 0 = delayed input, 1 = output */
	float        diff_min, diff_max;
	int          reset_delay;
} Chn;

/* minimal atom forge: append-only writer that keeps every open container's
 * size up to date (what lv2_atom_forge_* does in the SDK) */
typedef struct {
	uint8_t*  buf;
	uint32_t  cap, off;
	LV2_Atom* open[4];
	int       depth;
	bool      full;
} Forge;

typedef struct {
	const LV2_Atom_Sequence* control;
	LV2_Atom_Sequence*       notify;
	float*                   latency_port;

	uint32_t     n_chn;
	double       rate;
	uint32_t     latency;
	int          run_error_seen; /* the first failing run() is reported on stderr, the others are silent (and silence) */
	prot_engine* engine;

	uint32_t hold_time, frames_per_period;
	float    falloff;

	LV2_URID_Map* map;
	Urids         u;
	Forge         forge;

	bool  ui_active, send_state;
	float ui_scale;
	bool  link;

	Chn chn[MAX_CHANNELS];
} Plugin;

/* ---------------------------------------------------------------- forge --- */
static void*
forge_append (Forge* f, const void* data, uint32_t size)
{
	if (f->full || f->off + size > f->cap) {
		f->full = true;
		return NULL;
	}
	void* at = f->buf + f->off;
	if (data) {
		memcpy (at, data, size);
	} else {
		memset (at, 0, size);
	}
	f->off += size;
	for (int i = 0; i < f->depth; ++i) {
		f->open[i]->size += size;
	}
	return at;
}

static void*
forge_write_padded (Forge* f, const void* data, uint32_t size)
{
	void* at = forge_append (f, data, size);
	if (at && prot_atom_pad (size) != size) {
		forge_append (f, NULL, prot_atom_pad (size) - size);
	}
	return at;
}

static void
forge_begin (Forge* f, LV2_Atom_Sequence* seq, LV2_URID type_sequence)
{
	f->buf   = (uint8_t*)seq;
	f->cap   = seq->atom.size; /* the host announces the capacity here, src/phaserotate.c:796 */
	f->off   = 0;
	f->depth = 0;
	f->full  = false;
	const LV2_Atom_Sequence head = { { (uint32_t)sizeof (LV2_Atom_Sequence_Body), type_sequence }, { 0, 0 } };
	LV2_Atom* a = (LV2_Atom*)forge_write_padded (f, &head, sizeof (head));
	if (a) {
		f->open[f->depth++] = a;
	}
}

static bool
forge_object_open (Forge* f, LV2_URID type_object, LV2_URID otype)
{
	const int64_t frame_time = 0;
	if (!forge_append (f, &frame_time, sizeof (frame_time))) {
		return false;
	}
	const LV2_Atom_Object obj = { { (uint32_t)sizeof (LV2_Atom_Object_Body), type_object }, { 1, otype } };
	LV2_Atom* a = (LV2_Atom*)forge_write_padded (f, &obj, sizeof (obj));
	if (!a || f->depth >= 4) {
		return false;
	}
	f->open[f->depth++] = a;
	return true;
}

static void
forge_object_close (Forge* f)
{
	if (f->depth > 1) {
		--f->depth;
	}
}

static void
forge_prop_head (Forge* f, LV2_URID key)
{
	const uint32_t kc[2] = { key, 0 };
	forge_append (f, kc, sizeof (kc));
}

static void
forge_prop_float (Forge* f, LV2_URID key, LV2_URID type_float, float v)
{
	forge_prop_head (f, key);
	const LV2_Atom_Float a = { { sizeof (float), type_float }, v };
	forge_write_padded (f, &a, sizeof (a));
}

static void
forge_prop_int (Forge* f, LV2_URID key, LV2_URID type, int32_t v)
{
	forge_prop_head (f, key);
	const LV2_Atom_Int a = { { sizeof (int32_t), type }, v };
	forge_write_padded (f, &a, sizeof (a));
}

/* --------------------------------------------------------------- meters --- */
/* hold 0.5 s, then fall off 15 dB/s; absolute peak kept (src/phaserotate.c:451-470) */
static float
meter_feed (const Plugin* self, Meter* m, float level)
{
	if (!isfinite (level)) {
		level = 0;
	}
	if (level > m->peak) {
		m->peak = level;
	}
	if (level > m->momentary) {
		m->momentary = level;
		m->hold      = (int)self->hold_time;
	} else if (m->hold > 0) {
		m->hold -= (int)self->frames_per_period;
	} else {
		m->momentary *= self->falloff;
		m->momentary += 1e-20f;
	}
	return level;
}

static float
abs_max (const float* d, uint32_t n, float start)
{
	float pk = start;
	for (uint32_t i = 0; i < n; ++i) {
		const float v = fabsf (d[i]);
		if (v > pk) {
			pk = v;
		}
	}
	return pk;
}

static void
meters_clear (Chn* c) /* src/phaserotate.c:489-495 */
{
	c->mtr[0].peak = c->mtr[1].peak = 0;
	c->mtr[0].momentary = c->mtr[1].momentary = 0;
	c->diff_min = c->diff_max = 1;
}

/* input meter on the latency-delayed input (src/phaserotate.c:573-609); the
 * delay line update reproduces the reference's behaviour for short buffers
 * (latency > 2 * n_samples: only slot n_samples is refreshed) so
 * that the GUI sees the same numbers */
static float
meter_delayed_input (Plugin* self, Chn* c, const float* io, uint32_t n)
{
	const uint32_t L = self->latency;
	float          lvl;
	if (n < L) {
		lvl = meter_feed (self, &c->mtr[0], abs_max (c->dly, n, 0));
		const uint32_t keep = L - n;
		if (keep <= n) {
			memcpy (c->dly, c->dly + n, keep * sizeof (float));
		} else {
			memcpy (c->dly, c->dly + n, n * sizeof (float));
			c->dly[n] = c->dly[keep - 1]; /* net effect of src/phaserotate.c:586-588 */
		}
		memcpy (c->dly + keep, io, n * sizeof (float));
	} else {
		float pk = abs_max (c->dly, L, 0);
		pk       = abs_max (io, n - L, pk);
		memcpy (c->dly, io + (n - L), L * sizeof (float));
		lvl = meter_feed (self, &c->mtr[0], pk);
	}
	return lvl;
}

/* ------------------------------------------------------------ LV2 calls --- */
static void
cleanup (LV2_Handle instance)
{
	Plugin* self = (Plugin*)instance;
	if (!self) {
		return;
	}
	for (uint32_t c = 0; c < MAX_CHANNELS; ++c) {
		free (self->chn[c].dly);
	}
	if (self->engine) { /* deferred run(): periods the device was late for went out as silence -- say how many, if any */
		uint64_t late = 0, dropped = 0;
		if (prot_lv2_deferred_stats (self->engine, &late, &dropped) == PROT_OK && late + dropped > 0) {
			fprintf (stderr, "phaserotate.lv2: deferred run(): %llu period(s) late, %llu not posted (silence went out instead)\n",
			         (unsigned long long)late, (unsigned long long)dropped);
		}
	}
	prot_engine_destroy (self->engine);
	free (self);
}

static LV2_Handle
instantiate (const LV2_Descriptor* descriptor, double rate, const char* bundle_path,
             const LV2_Feature* const* features)
{
	Plugin* self = (Plugin*)calloc (1, sizeof (Plugin));
	if (!self) {
		return NULL;
	}
	if (!strcmp (descriptor->URI, PHASEROTATE_URI)) { /* src/phaserotate.c:233-240 */
		self->n_chn = 1;
	} else if (!strcmp (descriptor->URI, PHASEROTATE_URI "#stereo")) {
		self->n_chn = 2;
	} else {
		free (self);
		return NULL;
	}

	const LV2_Options_Option* options = NULL;
	for (int i = 0; features && features[i]; ++i) {
		if (!strcmp (features[i]->URI, LV2_URID__map)) {
			self->map = (LV2_URID_Map*)features[i]->data;
		} else if (!strcmp (features[i]->URI, LV2_OPTIONS__options)) {
			options = (const LV2_Options_Option*)features[i]->data;
		}
	}
	if (!self->map) { /* src/phaserotate.c:252-256 */
		fprintf (stderr, "phaserotate.lv2 error: Host does not support urid:map\n");
		free (self);
		return NULL;
	}

	LV2_URID_Map* m = self->map;
#define MAP(field, uri) self->u.field = m->map (m->handle, uri)
	MAP (atom_Blank, LV2_ATOM__Blank);
	MAP (atom_Object, LV2_ATOM__Object);
	MAP (atom_Float, LV2_ATOM__Float);
	MAP (atom_Int, LV2_ATOM__Int);
	MAP (atom_Bool, LV2_ATOM__Bool);
	MAP (atom_Sequence, LV2_ATOM__Sequence);
	MAP (ui_on, PROT_URI "ui_on");
	MAP (ui_off, PROT_URI "ui_off");
	MAP (reset_peaks, PROT_URI "reset_peaks");
	MAP (state, PROT_URI "state");
	MAP (s_uiscale, PROT_URI "uiscale");
	MAP (s_link, PROT_URI "link");
	MAP (levels, PROT_URI "levels");
	MAP (l_channel, PROT_URI "l_channel");
	MAP (l_in_cur, PROT_URI "l_in_cur");
	MAP (l_in_mom, PROT_URI "l_in_mom");
	MAP (l_in_peak, PROT_URI "l_in_peak");
	MAP (l_out_cur, PROT_URI "l_out_cur");
	MAP (l_out_mom, PROT_URI "l_out_mom");
	MAP (l_out_peak, PROT_URI "l_out_peak");
	MAP (l_diff_cur, PROT_URI "l_diff_cur");
	MAP (l_diff_min, PROT_URI "l_diff_min");
	MAP (l_diff_max, PROT_URI "l_diff_max");
#undef MAP

	self->ui_scale = 1.0f; /* src/phaserotate.c:299-301 */
	uint32_t max_block = 0;
	if (options) {         /* src/phaserotate.c:261-276 (the reference then overwrites it with 1.0 at :300) */
		const LV2_URID key = m->map (m->handle, "http://lv2plug.in/ns/extensions/ui#scaleFactor");
		/* buf-size:maxBlockLength: the longest run() the host will ask for -- the engine reserves its staging buffers
		 * for it here, so that run() never allocates (longer calls would be cut into reserved-size pieces) */
		const LV2_URID kmb = m->map (m->handle, "http://lv2plug.in/ns/ext/buf-size#maxBlockLength");
		for (const LV2_Options_Option* o = options; o->key; ++o) {
			if (o->context == LV2_OPTIONS_INSTANCE && o->key == key && o->type == self->u.atom_Float) {
				(void)*(const float*)o->value; /* parsed and, as in the reference, not kept */
			}
			if (o->context == LV2_OPTIONS_INSTANCE && o->key == kmb && o->type == self->u.atom_Int && o->value) {
				const int32_t v = *(const int32_t*)o->value;
				max_block       = v > 0 ? (uint32_t)v : 0;
			}
		}
	}

	self->rate      = rate;
	self->hold_time = (uint32_t)(0.5 * rate + 0.5f); /* src/phaserotate.c:303 */

	prot_config cfg;
	memset (&cfg, 0, sizeof (cfg));
	cfg.kind   = PROT_KIND_LV2;
	cfg.param  = rate;
	cfg.n_chn  = self->n_chn;
	cfg.device = getenv ("PROT_DEVICE") ? atoi (getenv ("PROT_DEVICE")) : 0;
	int status   = 0;
	self->engine = prot_engine_create (&cfg, &status);
	if (!self->engine) {
		fprintf (stderr, "phaserotate.lv2 error: cannot create the GPU engine: %s (%s)\n",
		         prot_strerror (status), prot_last_error ());
		cleanup ((LV2_Handle)self);
		return NULL;
	}
	if (max_block > 8192 && prot_lv2_set_max_block (self->engine, max_block) != PROT_OK) { /* (8192 frames are reserved by default) */
		fprintf (stderr, "phaserotate.lv2 error: cannot reserve buffers for %u-frame blocks: %s\n", max_block, prot_last_error ());
		cleanup ((LV2_Handle)self);
		return NULL;
	}
	prot_info info;
	prot_engine_info (self->engine, &info);
	self->latency = info.latency; /* 1792 / 2560 / 5120, src/phaserotate.c:297 */

	/* Deferred run() (opt-in): run() posts its period to the device and returns the period before -- it never waits, which is what
	 * lets a bundle declare lv2:hardRTCapable (lv2ttl/phaserotate.ports.in:7) -- at the price of D more frames of latency, reported
	 * on the latency port like the rest (:788).  Asked for by the bundle (a file `deferred` beside the binary, written by
	 * make_bundle.py --deferred N) or by the environment (PROT_LV2_DEFERRED=N); N = frames, 1 = the host's maxBlockLength (1024 if
	 * the host does not say). */
	uint32_t deferred = 0;
	if (getenv ("PROT_LV2_DEFERRED")) {
		deferred = (uint32_t)strtoul (getenv ("PROT_LV2_DEFERRED"), NULL, 10);
	} else if (bundle_path && strlen (bundle_path) < 3900) {
		char path[4096];
		snprintf (path, sizeof (path), "%s%sdeferred", bundle_path, bundle_path[0] && bundle_path[strlen (bundle_path) - 1] == '/' ? "" : "/");
		FILE* f = fopen (path, "r");
		if (f) {
			unsigned v = 0;
			if (fscanf (f, "%u", &v) == 1) {
				deferred = v;
			}
			fclose (f);
		}
	}
	if (deferred == 1) {
		deferred = max_block > 0 ? max_block : 1024;
	}
	if (deferred > 0) {
		if (prot_lv2_set_deferred (self->engine, deferred) == PROT_OK) {
			self->latency += deferred;
		} else {
			fprintf (stderr, "phaserotate.lv2 warning: deferred run() not available (%s): run() waits for the device\n", prot_last_error ());
		}
	}

	for (uint32_t c = 0; c < self->n_chn; ++c) {
		self->chn[c].dly = (float*)calloc (self->latency, sizeof (float));
		if (!self->chn[c].dly) {
			cleanup ((LV2_Handle)self);
			return NULL;
		}
		self->chn[c].diff_min = self->chn[c].diff_max = 1;
	}
	return (LV2_Handle)self;
}

static void
connect_port (LV2_Handle instance, uint32_t port, void* data)
{
	Plugin* self = (Plugin*)instance;
	switch (port) {
		case PORT_ATOM_CONTROL: self->control = (const LV2_Atom_Sequence*)data; return;
		case PORT_ATOM_NOTIFY: self->notify = (LV2_Atom_Sequence*)data; return;
		case PORT_LATENCY: self->latency_port = (float*)data; return;
		default: break;
	}
	const uint32_t idx = port - PORT_ANGLE0; /* 3 ports per channel: angle, in, out */
	const uint32_t c   = idx / 3;
	if (c >= MAX_CHANNELS) {
		return;
	}
	switch (idx % 3) {
		case 0: self->chn[c].angle = (const float*)data; break;
		case 1: self->chn[c].in = (const float*)data; break;
		case 2: self->chn[c].out = (float*)data; break;
	}
}

static void
activate (LV2_Handle instance)
{
	Plugin* self = (Plugin*)instance;
	prot_engine_reset (self->engine); /* channel_reset, src/phaserotate.c:169-177 */
	for (uint32_t c = 0; c < self->n_chn; ++c) {
		meters_clear (&self->chn[c]);
		self->chn[c].reset_delay = (int)self->latency; /* :518 */
	}
}

static const LV2_Atom*
object_get (const LV2_Atom_Object* obj, LV2_URID key)
{
	const uint8_t* p   = (const uint8_t*)(obj + 1);
	const uint8_t* end = (const uint8_t*)&obj->body + obj->atom.size;
	while (p + sizeof (LV2_Atom_Property_Body) <= end) {
		const LV2_Atom_Property_Body* prop = (const LV2_Atom_Property_Body*)p;
		if (prop->key == key) {
			return &prop->value;
		}
		p += prot_atom_pad ((uint32_t)sizeof (LV2_Atom_Property_Body) + prop->value.size);
	}
	return NULL;
}

static void
read_control (Plugin* self) /* src/phaserotate.c:800-830 */
{
	const LV2_Atom_Sequence* seq = self->control;
	const uint8_t*           p   = (const uint8_t*)(&seq->body + 1);
	const uint8_t*           end = (const uint8_t*)&seq->body + seq->atom.size;
	while (p + sizeof (LV2_Atom_Event) <= end) {
		const LV2_Atom_Event* ev = (const LV2_Atom_Event*)p;
		if (ev->body.type == self->u.atom_Blank || ev->body.type == self->u.atom_Object) {
			const LV2_Atom_Object* obj = (const LV2_Atom_Object*)&ev->body;
			const LV2_URID         ot  = obj->body.otype;
			if (ot == self->u.ui_off) {
				self->ui_active = false;
			} else if (ot == self->u.ui_on) {
				self->ui_active  = true;
				self->send_state = true;
			} else if (ot == self->u.reset_peaks) {
				for (uint32_t c = 0; c < self->n_chn; ++c) {
					meters_clear (&self->chn[c]);
				}
			} else if (ot == self->u.state) {
				const LV2_Atom* v = object_get (obj, self->u.s_uiscale);
				if (v) {
					self->ui_scale = ((const LV2_Atom_Float*)v)->body;
				}
				v = object_get (obj, self->u.s_link);
				if (v) {
					self->link = ((const LV2_Atom_Bool*)v)->body;
				}
			}
		}
		p += prot_atom_pad ((uint32_t)sizeof (LV2_Atom_Event) + ev->body.size);
	}
}

static void
run (LV2_Handle instance, uint32_t n_samples)
{
	Plugin* self = (Plugin*)instance;

	for (uint32_t c = 0; c < self->n_chn; ++c) { /* forward non-in-place buffers, :779-785 */
		if (self->chn[c].in != self->chn[c].out) {
			memcpy (self->chn[c].out, self->chn[c].in, sizeof (float) * n_samples);
		}
	}
	*self->latency_port = (float)self->latency; /* :788 */

	if (!self->control || !self->notify) {
		return; /* latency measurement callback: no DSP, :790-793 */
	}

	forge_begin (&self->forge, self->notify, self->u.atom_Sequence);
	read_control (self);

	if (self->frames_per_period != n_samples) { /* meter fall-off per period: 15 dB/s, :832-838 */
		const float tme         = (float)n_samples / (float)self->rate;
		self->falloff           = powf (10.0f, -0.05f * 15.0f * tme);
		self->frames_per_period = n_samples;
	}

	float        lvl_in[MAX_CHANNELS];
	float        ang[MAX_CHANNELS];
	const float* inp[MAX_CHANNELS];
	float*       outp[MAX_CHANNELS];
	for (uint32_t c = 0; c < self->n_chn; ++c) {
		Chn* ch   = &self->chn[c];
		lvl_in[c] = meter_delayed_input (self, ch, ch->out, n_samples);

		/* a pending angle change re-arms the delayed reset of the ratio meters, :611, :497-509 */
		float target = *ch->angle / -360.f;
		target       = target < -.5f ? -.5f : (target > .5f ? .5f : target);
		float cur    = 0;
		prot_lv2_get_angle (self->engine, c, &cur);
		if (ch->reset_delay > 0) {
			ch->diff_min = ch->diff_max = 1;
			ch->mtr[1].momentary        = 0;
			ch->reset_delay -= (int)n_samples;
		}
		if (target != cur) {
			ch->reset_delay = (int)(self->latency + n_samples);
		}
		ang[c]  = *ch->angle;
		inp[c]  = ch->out; /* processing is in place on the output buffer, :563 */
		outp[c] = ch->out;
	}

	/* the hot path: one GPU call for all channels of this period */
	if (n_samples > 0 && prot_lv2_run (self->engine, inp, outp, n_samples, ang) != PROT_OK) {
		/* run() cannot report errors (no such path in LV2): emit silence rather than stale data (and say so once) */
		if (!self->run_error_seen) {
			self->run_error_seen = 1;
			fprintf (stderr, "phaserotate.lv2 error: run() failed: %s\n", prot_last_error ());
		}
		for (uint32_t c = 0; c < self->n_chn; ++c) {
			memset (self->chn[c].out, 0, sizeof (float) * n_samples);
		}
	}

	for (uint32_t c = 0; c < self->n_chn; ++c) { /* output meter + in/out ratio, :727-739 */
		Chn*  ch       = &self->chn[c];
		float lvl_out  = meter_feed (self, &ch->mtr[1], abs_max (ch->out, n_samples, 0));
		float lvl_diff = 1.0f;
		if (ch->mtr[0].momentary > 0.001f && ch->mtr[1].momentary > 0.001f) {
			lvl_diff = ch->mtr[1].momentary / ch->mtr[0].momentary;
			if (lvl_diff < ch->diff_min) {
				ch->diff_min = lvl_diff;
			}
			if (lvl_diff > ch->diff_max) {
				ch->diff_max = lvl_diff;
			}
		}
		if (self->ui_active && forge_object_open (&self->forge, self->u.atom_Object, self->u.levels)) { /* :741-771 */
			Forge* f = &self->forge;
			forge_prop_int (f, self->u.l_channel, self->u.atom_Int, (int32_t)c);
			forge_prop_float (f, self->u.l_in_cur, self->u.atom_Float, lvl_in[c]);
			forge_prop_float (f, self->u.l_in_mom, self->u.atom_Float, ch->mtr[0].momentary);
			forge_prop_float (f, self->u.l_in_peak, self->u.atom_Float, ch->mtr[0].peak);
			forge_prop_float (f, self->u.l_out_cur, self->u.atom_Float, lvl_out);
			forge_prop_float (f, self->u.l_out_mom, self->u.atom_Float, ch->mtr[1].momentary);
			forge_prop_float (f, self->u.l_out_peak, self->u.atom_Float, ch->mtr[1].peak);
			forge_prop_float (f, self->u.l_diff_cur, self->u.atom_Float, lvl_diff);
			forge_prop_float (f, self->u.l_diff_min, self->u.atom_Float, ch->diff_min);
			forge_prop_float (f, self->u.l_diff_max, self->u.atom_Float, ch->diff_max);
			forge_object_close (f);
		}
	}

	if (self->ui_active && self->send_state) { /* tx_state, :522-536, :845-848 */
		self->send_state = false;
		if (forge_object_open (&self->forge, self->u.atom_Object, self->u.state)) {
			forge_prop_float (&self->forge, self->u.s_uiscale, self->u.atom_Float, self->ui_scale);
			forge_prop_int (&self->forge, self->u.s_link, self->u.atom_Bool, self->link ? 1 : 0);
			forge_object_close (&self->forge);
		}
	}
}

static const void*
extension_data (const char* uri)
{
	(void)uri;
	return NULL; /* src/phaserotate.c:854-858 */
}

static LV2_Descriptor descriptor = {
	PHASEROTATE_URI, instantiate, connect_port, activate, run, NULL /* deactivate */, cleanup, extension_data
};

/* the one exported symbol, src/phaserotate.c:872-893 */
__attribute__ ((visibility ("default"))) const LV2_Descriptor*
lv2_descriptor (uint32_t index)
{
	switch (index) {
		case 0: descriptor.URI = PHASEROTATE_URI; return &descriptor;
		case 1: descriptor.URI = PHASEROTATE_URI "#stereo"; return &descriptor;
		default: return NULL;
	}
}
