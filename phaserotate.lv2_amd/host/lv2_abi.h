/*
 * lv2_abi.h -- the slice of the public LV2 C ABI this plugin needs, declared
 * from the LV2 specification (struct layouts and URIs are ABI facts; the LV2
 * SDK headers are not installed in this image).  Binary-compatible with
 * lv2/core/lv2.h, lv2/urid/urid.h, lv2/options/options.h and lv2/atom/atom.h,
 * so a real LV2 host can load the resulting phaserotate.so unchanged.
 */
#ifndef PROT_LV2_ABI_H
#define PROT_LV2_ABI_H

#include <stdbool.h>
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- core ---------------------------------------------------------------- */
typedef void* LV2_Handle;

typedef struct {
	const char* URI;
	void*       data;
} LV2_Feature;

typedef struct LV2_Descriptor {
	const char* URI;
	LV2_Handle (*instantiate) (const struct LV2_Descriptor* descriptor, double sample_rate,
	                           const char* bundle_path, const LV2_Feature* const* features);
	void (*connect_port) (LV2_Handle instance, uint32_t port, void* data_location);
	void (*activate) (LV2_Handle instance);
	void (*run) (LV2_Handle instance, uint32_t sample_count);
	void (*deactivate) (LV2_Handle instance);
	void (*cleanup) (LV2_Handle instance);
	const void* (*extension_data) (const char* uri);
} LV2_Descriptor;

/* ---- urid ---------------------------------------------------------------- */
#define LV2_URID__map "http://lv2plug.in/ns/ext/urid#map"
typedef uint32_t LV2_URID;
typedef void*    LV2_URID_Map_Handle;
typedef struct {
	LV2_URID_Map_Handle handle;
	LV2_URID (*map) (LV2_URID_Map_Handle handle, const char* uri);
} LV2_URID_Map;

/* ---- options ------------------------------------------------------------- */
#define LV2_OPTIONS__options "http://lv2plug.in/ns/ext/options#options"
typedef enum {
	LV2_OPTIONS_INSTANCE,
	LV2_OPTIONS_RESOURCE,
	LV2_OPTIONS_BLANK,
	LV2_OPTIONS_PORT
} LV2_Options_Context;

typedef struct {
	LV2_Options_Context context;
	uint32_t            subject;
	LV2_URID            key;
	uint32_t            size;
	LV2_URID            type;
	const void*         value;
} LV2_Options_Option;

/* ---- atoms --------------------------------------------------------------- */
#define LV2_ATOM_URI "http://lv2plug.in/ns/ext/atom"
#define LV2_ATOM__Blank LV2_ATOM_URI "#Blank"
#define LV2_ATOM__Bool LV2_ATOM_URI "#Bool"
#define LV2_ATOM__Float LV2_ATOM_URI "#Float"
#define LV2_ATOM__Int LV2_ATOM_URI "#Int"
#define LV2_ATOM__Object LV2_ATOM_URI "#Object"
#define LV2_ATOM__Sequence LV2_ATOM_URI "#Sequence"
#define LV2_ATOM__Vector LV2_ATOM_URI "#Vector"
#define LV2_ATOM__eventTransfer LV2_ATOM_URI "#eventTransfer"

typedef struct {
	uint32_t size; /* of the body, in bytes */
	uint32_t type;
} LV2_Atom;

typedef struct {
	LV2_Atom atom;
	int32_t  body;
} LV2_Atom_Int;

typedef LV2_Atom_Int LV2_Atom_Bool;

typedef struct {
	LV2_Atom atom;
	float    body;
} LV2_Atom_Float;

typedef struct {
	uint32_t key;
	uint32_t context;
	LV2_Atom value; /* followed by the value body */
} LV2_Atom_Property_Body;

typedef struct {
	uint32_t id;
	uint32_t otype;
	/* followed by LV2_Atom_Property_Body, each padded to 8 bytes */
} LV2_Atom_Object_Body;

typedef struct {
	LV2_Atom             atom;
	LV2_Atom_Object_Body body;
} LV2_Atom_Object;

typedef struct {
	union {
		int64_t frames;
		double  beats;
	} time;
	LV2_Atom body; /* followed by the body contents */
} LV2_Atom_Event;

typedef struct {
	uint32_t unit;
	uint32_t pad;
	/* followed by LV2_Atom_Event, each padded to 8 bytes */
} LV2_Atom_Sequence_Body;

typedef struct {
	LV2_Atom               atom;
	LV2_Atom_Sequence_Body body;
} LV2_Atom_Sequence;

static inline uint32_t
prot_atom_pad (uint32_t size)
{
	return (size + 7u) & ~7u;
}

#ifdef __cplusplus
}
#endif
#endif
