// wavio.h -- minimal RIFF/WAVE reader/writer for the phase-rotate CLI.
//
// Stands where libsndfile stands in the reference (sf_open / sf_readf_float /
// sf_writef_float, cli/phase-rotate.cc:683-702, :565-587, :950-1003): PCM
// 16/24/32-bit and IEEE float 32/64-bit, plain or WAVE_FORMAT_EXTENSIBLE,
// any channel count.  The output file takes the input's format (the reference
// copies SF_INFO, :697).  Sample conversion follows libsndfile's defaults:
// integer PCM <-> float by 2^(bits-1) on read and (2^(bits-1) - 1) on write.
// copy_metadata (cli/phase-rotate.cc:541-563: strings, cue points, broadcast info) becomes: the input's
// "LIST" (INFO strings), "cue " and "bext" chunks are carried over verbatim, in front of the samples.
// What libsndfile offers beyond that -- other containers (AIFF, FLAC, W64, ...), RF64, other sample encodings --
// is refused with an error text, never guessed.
// The whole file is decoded at once (the device passes work on whole files): straight from a read-only mapping, by up
// to 16 threads; the writer encodes 48 MiB at a time the same way.  Per-sample arithmetic is that of the scalar loop.
#pragma once
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <functional>
#include <memory>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <string>
#include <thread>
#include <vector>

namespace wavio {

// sample storage that is NOT zero-filled on resize (every element is written by the decoder; zero-filling 100 M floats
// costs as much as decoding them)
template <class T>
struct no_init_alloc : std::allocator<T> {
	template <class U> struct rebind { typedef no_init_alloc<U> other; };
	template <class U> void construct (U* p) { ::new ((void*)p) U; }
	template <class U, class A0, class... A> void construct (U* p, A0&& a0, A&&... a) { ::new ((void*)p) U (std::forward<A0> (a0), std::forward<A> (a)...); }
};
typedef std::vector<float, no_init_alloc<float>> Samples;

struct Chunk {
	char                 id[4];
	std::vector<uint8_t> body;
};

struct Info {
	uint32_t samplerate = 0;
	uint16_t channels   = 0;
	uint16_t bits       = 0;     // container bits per sample
	bool     is_float   = false;
	uint64_t frames     = 0;
	std::vector<Chunk> meta;     // LIST / cue / bext chunks of the input, in file order (<= 1 MiB each, 16 in all)
};

inline uint32_t rd32 (const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint16_t rd16 (const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// Sample conversion is the CLI's largest host cost on long files (a 10-minute stereo 96 kHz file is 115 M samples): the
// chunks below are converted by `workers()` threads, each its own contiguous slice -- per-sample arithmetic unchanged.
inline unsigned workers (uint64_t n_samples)
{
	const unsigned hw = std::thread::hardware_concurrency ();
	const unsigned w  = std::min<unsigned> (hw ? hw : 1, 16);
	return n_samples < (1u << 18) ? 1 : w;
}
template <class F>
inline void parallel_slices (uint64_t n, F&& f) // f (begin, end)
{
	const unsigned w = workers (n);
	if (w <= 1) {
		f ((uint64_t)0, n);
		return;
	}
	std::vector<std::thread> th;
	for (unsigned i = 0; i < w; ++i) {
		const uint64_t b = n * i / w, e = n * (i + 1) / w;
		th.emplace_back ([&f, b, e] () { f (b, e); });
	}
	for (std::thread& t : th) {
		t.join ();
	}
}

// one sample of the data chunk -> float (libsndfile's defaults: integer PCM by 1 / 2^(bits-1))
inline float decode_sample (const uint8_t* p, bool is_float, uint32_t bps)
{
	if (is_float) {
		if (bps == 4) {
			float v;
			memcpy (&v, p, 4);
			return v;
		}
		double d;
		memcpy (&d, p, 8);
		return (float)d;
	}
	if (bps == 2) {
		return (float)(int16_t)rd16 (p) / 32768.f;
	}
	if (bps == 3) {
		const int32_t s = (int32_t)((uint32_t)p[0] << 8 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 24);
		return (float)s / 2147483648.f;
	}
	return (float)(int32_t)rd32 (p) / 2147483648.f;
}

// Reads the whole file as interleaved float32.  Returns "" on success, else an error text.
// `on_header`, if given, is called once the format and the frame count are known and before the samples are decoded
// (the CLI starts creating its GPU engine there, under the decode).
inline std::string read_file (const char* path, Info& nfo, Samples& data, const std::function<void (const Info&)>& on_header = nullptr)
{
	FILE* f = fopen (path, "rb");
	if (!f) {
		return std::string ("System error : ") + strerror (errno) + ".";
	}
	uint8_t hdr[12];
	if (fread (hdr, 1, 12, f) != 12 || memcmp (hdr, "RIFF", 4) || memcmp (hdr + 8, "WAVE", 4)) {
		fclose (f);
		return "File contains data in an unknown format.";
	}
	bool     have_fmt = false;
	uint16_t tag      = 0;
	for (;;) {
		uint8_t ch[8];
		if (fread (ch, 1, 8, f) != 8) {
			break;
		}
		uint32_t sz = rd32 (ch + 4);
		if (!memcmp (ch, "fmt ", 4)) {
			if (sz < 16 || sz > 4096) { // (plain 16, extensible 40: a crafted size must not become a 4 GiB allocation)
				break;
			}
			std::vector<uint8_t> b (sz);
			if (fread (b.data (), 1, sz, f) != sz) {
				break;
			}
			tag            = rd16 (&b[0]);
			nfo.channels   = rd16 (&b[2]);
			nfo.samplerate = rd32 (&b[4]);
			nfo.bits       = rd16 (&b[14]);
			if (tag == 0xFFFE && sz >= 26) {
				tag = rd16 (&b[24]); // sub-format GUID starts with the plain tag
			}
			nfo.is_float = (tag == 3);
			have_fmt     = (tag == 1 || tag == 3) && nfo.channels > 0;
			if (sz & 1) {
				fseek (f, 1, SEEK_CUR);
			}
		} else if (!memcmp (ch, "data", 4)) {
			if (!have_fmt) {
				break;
			}
			const uint32_t bps = nfo.bits / 8;
			if (!((nfo.is_float && (bps == 4 || bps == 8)) || (!nfo.is_float && (bps == 2 || bps == 3 || bps == 4)))) {
				fclose (f);
				return "Unsupported sample format.";
			}
			long here = ftell (f);
			fseek (f, 0, SEEK_END);
			long end = ftell (f);
			fseek (f, here, SEEK_SET);
			uint64_t avail = (uint64_t)(end - here);
			uint64_t bytes = sz == 0xFFFFFFFFu || sz > avail ? avail : sz;
			nfo.frames     = bytes / ((uint64_t)bps * nfo.channels);
			const uint64_t n = nfo.frames * nfo.channels;
			if (on_header) {
				on_header (nfo);
			}
			data.resize (n);
			const bool is_float = nfo.is_float;
			// the whole data chunk mapped and decoded in place by all workers (no copy through a staging buffer) ...
			{
				const int fd = open (path, O_RDONLY);
				void*     mp = fd >= 0 ? mmap (nullptr, (size_t)end, PROT_READ, MAP_PRIVATE, fd, 0) : MAP_FAILED;
				if (fd >= 0) {
					close (fd);
				}
				if (mp != MAP_FAILED) {
					float* const         dst = data.data ();
					const uint8_t* const src = (const uint8_t*)mp + here;
					parallel_slices (n, [=] (uint64_t b, uint64_t e) {
						const uint8_t* p = src + b * bps;
						for (uint64_t i = b; i < e; ++i, p += bps) {
							dst[i] = decode_sample (p, is_float, bps);
						}
					});
					munmap (mp, (size_t)end);
					fseek (f, here + (long)(n * bps), SEEK_SET);
					goto decoded;
				}
			}
			// ... or, where the file cannot be mapped, read 48 MiB at a time
			{
			std::vector<uint8_t> raw ((size_t)48 << 20); // (a multiple of every sample size)
			uint64_t done = 0;
			while (done < n) {
				const uint64_t want = std::min<uint64_t> ((raw.size () / bps), n - done);
				if (fread (raw.data (), bps, want, f) != want) {
					fclose (f);
					return "Unexpected end of file.";
				}
				float* const         dst = data.data () + done;
				const uint8_t* const src = raw.data ();
				parallel_slices (want, [=] (uint64_t b, uint64_t e) {
					const uint8_t* p = src + b * bps;
					for (uint64_t i = b; i < e; ++i, p += bps) {
						dst[i] = decode_sample (p, is_float, bps);
					}
				});
				done += want;
			}
			}
decoded:
			// chunks behind the samples (where many writers put LIST / cue / bext)
			if (sz != 0xFFFFFFFFu && sz <= avail) {
				fseek (f, here + (long)sz + (long)(sz & 1), SEEK_SET);
				for (;;) {
					uint8_t c2[8];
					if (fread (c2, 1, 8, f) != 8) {
						break;
					}
					const uint32_t s2 = rd32 (c2 + 4);
					if ((!memcmp (c2, "LIST", 4) || !memcmp (c2, "cue ", 4) || !memcmp (c2, "bext", 4)) && s2 <= (1u << 20) && nfo.meta.size () < 16) {
						Chunk c;
						memcpy (c.id, c2, 4);
						c.body.resize (s2);
						if (fread (c.body.data (), 1, s2, f) != s2) {
							break;
						}
						nfo.meta.push_back (std::move (c));
						if (s2 & 1) {
							fseek (f, 1, SEEK_CUR);
						}
					} else if (fseek (f, (long)(s2 + (s2 & 1)), SEEK_CUR) != 0) {
						break;
					}
				}
			}
			fclose (f);
			return "";
		} else if ((!memcmp (ch, "LIST", 4) || !memcmp (ch, "cue ", 4) || !memcmp (ch, "bext", 4)) && sz <= (1u << 20) && nfo.meta.size () < 16) {
			Chunk c;
			memcpy (c.id, ch, 4);
			c.body.resize (sz);
			if (fread (c.body.data (), 1, sz, f) != sz) {
				break;
			}
			nfo.meta.push_back (std::move (c));
			if (sz & 1) {
				fseek (f, 1, SEEK_CUR);
			}
		} else {
			fseek (f, (long)(sz + (sz & 1)), SEEK_CUR);
		}
	}
	fclose (f);
	return "File contains data in an unknown format.";
}

class Writer
{
public:
	// "" on success
	std::string open (const char* path, const Info& nfo)
	{
		_nfo = nfo;
		_f   = fopen (path, "wb");
		if (!_f) {
			return std::string ("System error : ") + strerror (errno) + ".";
		}
		// WAVE_FORMAT_EXTENSIBLE where the plain header is not defined: more than two channels or more than 16 bits of PCM
		_ext = _nfo.channels > 2 || (!_nfo.is_float && _nfo.bits > 16);
		std::vector<uint8_t> h (header_size (), 0);
		fwrite (h.data (), 1, h.size (), _f); // patched in close()
		return "";
	}

	// returns frames written
	uint64_t writef (const float* buf, uint64_t frames)
	{
		const uint32_t bps      = _nfo.bits / 8;
		const uint32_t chn      = _nfo.channels;
		const bool     is_float = _nfo.is_float;
		const uint64_t step     = std::max<uint64_t> (1, ((uint64_t)48 << 20) / ((uint64_t)bps * chn)); // frames per 48 MiB of output
		std::vector<uint8_t> raw ((size_t)std::min<uint64_t> (frames, step) * bps * chn);
		uint64_t done = 0;
		while (done < frames) {
			const uint64_t nf = std::min<uint64_t> (step, frames - done), n = nf * chn;
			const float* const src = buf + done * chn;
			uint8_t* const     dst = raw.data ();
			parallel_slices (n, [=] (uint64_t b0, uint64_t e0) {
				uint8_t* p = dst + b0 * bps;
				for (uint64_t i = b0; i < e0; ++i, p += bps) {
					const float v = src[i];
					if (is_float) {
						if (bps == 4) {
							memcpy (p, &v, 4);
						} else {
							const double d = v;
							memcpy (p, &d, 8);
						}
					} else {
						double  scale = bps == 2 ? 32767.0 : (bps == 3 ? 8388607.0 : 2147483647.0);
						double  s     = std::nearbyint ((double)v * scale);
						s             = std::min (std::max (s, -scale - 1.0), scale);
						int32_t q     = (int32_t)s;
						for (uint32_t b = 0; b < bps; ++b) {
							p[b] = (uint8_t)((uint32_t)q >> (8 * b));
						}
					}
				}
			});
			const uint64_t w = fwrite (raw.data (), (size_t)bps * chn, nf, _f);
			_frames += w;
			done += w;
			if (w != nf) {
				break;
			}
		}
		return done;
	}

	void close ()
	{
		if (!_f) {
			return;
		}
		const uint32_t bps  = _nfo.bits / 8;
		const uint64_t data = _frames * bps * _nfo.channels;
		if (data & 1) {
			fputc (0, _f);
		}
		std::vector<uint8_t> h (header_size (), 0);
		auto w32 = [] (uint8_t* p, uint32_t v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; };
		auto w16 = [] (uint8_t* p, uint16_t v) { p[0] = v; p[1] = v >> 8; };
		uint8_t* p = h.data ();
		memcpy (p, "RIFF", 4);
		w32 (p + 4, (uint32_t)std::min<uint64_t> (h.size () - 8 + data + (data & 1), 0xFFFFFFFFu));
		memcpy (p + 8, "WAVEfmt ", 8);
		const uint16_t tag = _nfo.is_float ? 3 : 1;
		w32 (p + 16, _ext ? 40 : 16);
		w16 (p + 20, _ext ? 0xFFFE : tag);
		w16 (p + 22, _nfo.channels);
		w32 (p + 24, _nfo.samplerate);
		w32 (p + 28, _nfo.samplerate * bps * _nfo.channels);
		w16 (p + 32, (uint16_t)(bps * _nfo.channels));
		w16 (p + 34, _nfo.bits);
		p += 36;
		if (_ext) {
			static const uint8_t guid_tail[14] = { 0x00, 0x00, 0x00, 0x00, 0x10, 0x00, 0x80, 0x00, 0x00, 0xAA, 0x00, 0x38, 0x9B, 0x71 };
			w16 (p, 22);          // cbSize
			w16 (p + 2, _nfo.bits); // valid bits
			w32 (p + 4, 0);       // channel mask: unspecified
			w16 (p + 8, tag);     // sub-format GUID = tag + the fixed tail
			memcpy (p + 10, guid_tail, 14);
			p += 24;
		}
		for (const Chunk& c : _nfo.meta) { // copy_metadata, cli/phase-rotate.cc:541-563
			memcpy (p, c.id, 4);
			w32 (p + 4, (uint32_t)c.body.size ());
			memcpy (p + 8, c.body.data (), c.body.size ());
			p += 8 + c.body.size () + (c.body.size () & 1);
		}
		memcpy (p, "data", 4);
		w32 (p + 4, (uint32_t)std::min<uint64_t> (data, 0xFFFFFFFFu));
		fseek (_f, 0, SEEK_SET);
		fwrite (h.data (), 1, h.size (), _f);
		fclose (_f);
		_f = nullptr;
	}

	~Writer () { close (); }

private:
	size_t header_size () const
	{
		size_t n = 12 + 8 + (_ext ? 40 : 16) + 8;
		for (const Chunk& c : _nfo.meta) {
			n += 8 + c.body.size () + (c.body.size () & 1);
		}
		return n;
	}
	FILE*    _f = nullptr;
	Info     _nfo;
	bool     _ext    = false;
	uint64_t _frames = 0;
};

} // namespace wavio
