// CPU-only harness for host/wavio.h (no GPU, no libprot_hip): reads a WAV, writes it back in the same format.
// usage: wavio_roundtrip <in.wav> <out.wav>   -> prints "rate channels bits is_float frames meta_chunks"
#include "wavio.h"
int main (int argc, char** argv)
{
	if (argc != 3) {
		return 2;
	}
	wavio::Info    nfo;
	wavio::Samples data;
	const std::string err = wavio::read_file (argv[1], nfo, data);
	if (!err.empty ()) {
		fprintf (stderr, "%s\n", err.c_str ());
		return 1;
	}
	wavio::Writer w;
	if (!w.open (argv[2], nfo).empty ()) {
		return 1;
	}
	if (w.writef (data.data (), nfo.frames) != nfo.frames) {
		return 1;
	}
	w.close ();
	printf ("%u %u %u %d %llu %zu\n", nfo.samplerate, nfo.channels, nfo.bits, nfo.is_float ? 1 : 0, (unsigned long long)nfo.frames, nfo.meta.size ());
	return 0;
}
