// phase_rotate_cli.cc -- `phase-rotate`, the drop-in for the reference's
// command-line tool (cli/phase-rotate.cc) on MI355X.
//
// Same options, validation, messages, exit codes and result text; same analysis
// procedure (coarse pass -> candidate minima -> fine passes -> channel
// phase-distance unwrapping, cli/phase-rotate.cc:784-947) and apply semantics
// (:950-1003).  All DSP -- Hilbert FFT convolution, rotation, per-angle abs-max
// sweep -- runs in the gfx950 kernels behind the C ABI (include/prot_hip.h);
// this file is option parsing, file I/O and the selection logic on the reduced
// [n_chn][360] peak table.
//
// Differences, all outside the hot path: WAV only (own reader/writer instead of
// libsndfile, no metadata copy); the file is read once and kept in memory
// instead of being re-read for every fine pass (:872); a partial last block is
// applied correctly (the reference's tail is broken).  No CPU DSP
// fallback: without a GPU the tool exits with failure.
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <getopt.h>
#include <memory>
#include <thread>
#include <limits>
#include <map>
#include <string>
#include <vector>

#include "prot_hip.h"
#include "wavio.h"

#ifndef VERSION
#define VERSION "0.5-mi355x"
#endif

namespace {

constexpr int   kHalfDeg  = PROT_SUBSAMPLE;  // angle indices are half degrees
constexpr int   kAngles   = PROT_MAXSAMPLE;  // 360 indices = 180 degrees
constexpr float kInf      = std::numeric_limits<float>::infinity ();

[[noreturn]] void die (const char* msg)
{
	fputs (msg, stderr);
	::exit (EXIT_FAILURE);
}

void gpu_call (int rc, const char* what)
{
	if (rc != PROT_OK) {
		fprintf (stderr, "Error: %s failed: %s (%s)\n", what, prot_strerror (rc), prot_last_error ());
		::exit (EXIT_FAILURE);
	}
}

float to_dB (float coeff) // cli/phase-rotate.cc:76-83
{
	return coeff < 1e-15 ? -kInf : 20.0f * log10f (coeff);
}

// help2man-compatible text, word for word the reference's (cli/phase-rotate.cc:489-539)
// plus two lines about this build
const char* const kHelp =
    "phase-rotate - Audio File Phase Rotation Util.\n\n"
    "Usage: phase-rotate [ OPTIONS ] <file> [out-file]\n\n"
    "Options:\n"
    "  -a, --angle <n>[,<n>]*     specify phase angle to apply\n"
    "  -f, --fftlen <num>         process-block size, freq. resolution\n"
    "  -h, --help                 display this help and exit\n"
    "  -l, --link-channels        use downmixed mono peak for analysis\n"
    "  -s, --stride <num>         analysis step-size\n"
    "  -v, --verbose              show processing information\n"
    "  -V, --version              print version information and exit\n"
    "\n\n"
    "This utility analyzes the given audio file to find a phase-rotation\n"
    "angle that results in minimal digital-peak, while retaining overall\n"
    "sound and loudness.\n\n"
    "If both input and output file are given, the analysis results applied, and\n"
    "a new file with optimized phase is written. Otherwise the analysis results\n"
    "are only printed to standard output.\n\n"
    "Analysis is performed in two steps, first a coarse analysis is performed,\n"
    "calculating peak for angles distanced `stride' degrees apart. Then local\n"
    "minimums are explored in a second step.\n\n"
    "Verbose analysis allows to plot the digital peak vs phase-rotation.\n"
    "The output is in gnuplot(1) data file format.\n\n"
    "If the -a option is specified, no analysis is performed but the given,\n"
    "phase-angle(s) are directly applied. This requires both input and output\n"
    "files to be given. If a single angle is given it is applied to all channels\n"
    "of the file. Otherwise one has to specify the same number of phase-angles as\n"
    "there are channels in the file.\n\n\n"
    "Examples:\n"
    "phase-rotate -l my-music.wav out-file.wav\n\n"
    "phase-rotate -vv -s 3 my-music.wav\n\n"
    "phase-rotate -a 10,20 in.wav out.wav\n\n"
    "This build runs the DSP on an AMD MI355X (gfx950) through libprot_hip.\n"
    "Environment: PROT_DEVICE=<n> selects the HIP device; PROT_GPUS=<n> shards the\n"
    "analysis over n GPUs of the node (peak tables merged with an RCCL all-reduce).\n";

struct Options {
	const char*      in_path  = nullptr;
	const char*      out_path = nullptr;
	char*            angle_arg = nullptr;
	int              stride   = 12 * kHalfDeg;
	int              verbose  = 0;
	bool             link     = false;
	unsigned         blksiz   = 0;
};

Options parse_args (int argc, char** argv)
{
	static const struct option longopts[] = {
		{ "angle", required_argument, 0, 'a' },   { "fftlen", required_argument, 0, 'f' },
		{ "stride", required_argument, 0, 's' },  { "help", no_argument, 0, 'h' },
		{ "link-channels", no_argument, 0, 'l' }, { "version", no_argument, 0, 'V' },
		{ "verbose", no_argument, 0, 'v' },       { 0, 0, 0, 0 },
	};
	Options o;
	int     c;
	while ((c = getopt_long (argc, argv, "a:f:hls:Vv", longopts, nullptr)) != EOF) {
		switch (c) {
			case 'a': o.angle_arg = optarg; break;
			case 'f': o.blksiz = (unsigned)atoi (optarg); break;
			case 'l': o.link = true; break;
			case 's': o.stride = atoi (optarg); break;
			case 'v': ++o.verbose; break;
			case 'h':
				fputs (kHelp, stdout);
				::exit (EXIT_SUCCESS);
			case 'V':
				printf ("phase-rotate version %s\n\n", VERSION);
				printf ("MI355X/HIP implementation of x42 phase-rotate (GPL, Robin Gareus <robin@gareus.org>)\n");
				::exit (EXIT_SUCCESS);
			default: die ("Error: unrecognized option. See --help for usage information.\n");
		}
	}
	// validation order and texts of cli/phase-rotate.cc:663-681
	if (optind + 1 > argc) {
		die ("Error: Missing parameter. See --help for usage information.\n");
	}
	if (o.stride < 1 || o.stride > 45 * kHalfDeg || (kAngles % o.stride) != 0) {
		die ("Error: 180 deg is not evenly dividable by given stride.\n");
	}
	if (o.blksiz != 0 && (o.blksiz < 1024 || o.blksiz > 32768)) {
		die ("Error: fft-len is out of bounds; valid range 1024..32768\n");
	}
	if (o.angle_arg && optind + 2 > argc) {
		die ("Error: -a, --angle option requires an output file to be given.\n");
	}
	o.in_path = argv[optind];
	if (optind + 1 < argc) {
		o.out_path = argv[optind + 1];
	}
	return o;
}

// "-a n[,n]*" -> half-degree indices, one per channel (cli/phase-rotate.cc:718-740)
std::vector<int> parse_angles (char* arg, int n_channels)
{
	std::vector<int> idx;
	char*            save = nullptr;
	for (char* tok = strtok_r (arg, ",", &save); tok; tok = strtok_r (nullptr, ",", &save)) {
		char*        end = nullptr;
		const double deg = strtod (tok, &end);
		if (*end != '\0' || deg < -180 || deg > 180) {
			die ("Error: Invalid angle speficied, value needs to be -180 .. +180.\n");
		}
		idx.push_back ((int)round (deg * (float)kHalfDeg));
	}
	if (idx.size () == 1) {
		idx.resize ((size_t)n_channels, idx[0]);
	}
	if (idx.size () < (size_t)n_channels) {
		die ("Error: file has more channels than angles were specified.\n");
	}
	return idx;
}

// process block size: -f value or samplerate/8, next power of two, clamped (cli/phase-rotate.cc:749-755)
unsigned pick_block_size (unsigned requested, unsigned samplerate)
{
	unsigned want = (requested == 0 || requested > 32768) ? samplerate / 8 : requested;
	unsigned p2   = 2;
	while (p2 < want) {
		p2 <<= 1;
	}
	return std::min (32768u, std::max (1024u, p2));
}

// host copy of the engine's peak table with the reference's accessors (cli/phase-rotate.cc:275-299)
class Peaks
{
public:
	Peaks (prot_engine* e, int n_chn) : _n (n_chn), _v ((size_t)n_chn * kAngles)
	{
		gpu_call (prot_cli_peaks (e, _v.data ()), "prot_cli_peaks");
	}
	float mono (int a) const
	{
		float p = 0;
		for (int c = 0; c < _n; ++c) {
			p = std::max (p, _v[(size_t)c * kAngles + wrap (a)]);
		}
		return p;
	}
	float chn (int c, int a) const { return (c < 0 || c >= _n) ? mono (a) : _v[(size_t)c * kAngles + wrap (a)]; }

private:
	static int wrap (int a) { return ((a % kAngles) + kAngles) % kAngles; }
	int                _n;
	std::vector<float> _v;
};

void print_table_row (const Peaks& pk, int n_chn, int a)
{
	printf ("%.2f %.4f", (((a % kAngles) + kAngles) % kAngles) / (float)kHalfDeg, to_dB (pk.mono (a)));
	for (int c = 0; c < n_chn; ++c) {
		printf (" %.4f", to_dB (pk.chn (c, a)));
	}
	printf ("\n");
}

struct Result {
	std::vector<int>   angle;  // half-degree index, possibly negative after unwrapping
	std::vector<float> best;   // peak the search minimised (mono peak with -l), inf = "cannot find min."
	std::vector<float> at_zero, at_min;
};

// the minimum search of cli/phase-rotate.cc:784-929 on the whole file
// analysis pass over the whole file: one engine, or PROT_GPUS engines each sweeping a range of
// frame blocks followed by a MAX all-reduce of their peak tables
void analysis_pass (const std::vector<prot_engine*>& gpus, const wavio::Samples& file, uint64_t n_frames,
                    int a0, int a1, int step, int chn)
{
	if (gpus.size () == 1) {
		gpu_call (prot_cli_analyze_file (gpus[0], file.data (), n_frames, a0, a1, step, chn), "analysis");
	} else {
		gpu_call (prot_cli_analyze_file_multi (gpus.data (), (int)gpus.size (), file.data (), n_frames, a0, a1, step, chn),
		          "analysis");
	}
}

Result find_minimum (const std::vector<prot_engine*>& gpus, const wavio::Samples& file, uint64_t n_frames, int n_chn,
                     const Options& opt, FILE* vfd)
{
	prot_engine* pr = gpus[0]; // after a pass every engine holds the merged table
	const int stride = opt.stride;
	Result    r;
	r.angle.assign (n_chn, 0);
	r.best.assign (n_chn, kInf);
	r.at_zero.assign (n_chn, 0.f);
	r.at_min.assign (n_chn, 0.f);

	// coarse pass, all channels (:784)
	analysis_pass (gpus, file, n_frames, 0, kAngles, stride, -1);
	const Peaks coarse (pr, n_chn);
	if (opt.verbose > 1) { // gnuplot table (:800-813)
		printf ("# Angle mono-peak");
		for (int c = 0; c < n_chn; ++c) {
			printf (" chn-%d", c + 1);
		}
		printf ("\n");
		for (int a = 0; a < kAngles; a += stride) {
			print_table_row (coarse, n_chn, a);
		}
	}

	// candidates: coarse angles within 7 % of the channel's peak range above its minimum
	// (:823-857); an ordered map, so later passes visit them by ascending angle
	std::map<int, std::vector<int>> candidates;
	for (int c = 0; c < n_chn; ++c) {
		const int src = opt.link ? -1 : c;
		float     lo = kInf, hi = 0;
		r.at_zero[c] = coarse.chn (c, 0);
		for (int a = 0; a < kAngles; a += stride) {
			lo = std::min (lo, coarse.chn (src, a));
			hi = std::max (hi, coarse.chn (src, a));
		}
		float window = hi - lo;
		if (window == 0) {
			candidates[0].push_back (c); // flat: "cannot find min." unless a fine pass finds one
			continue;
		}
		if (stride > 1) {
			window *= .07;
		} else {
			window    = 0;
			r.best[c] = lo;
		}
		for (int a = 0; a < kAngles; a += stride) {
			const float p = coarse.chn (src, a);
			if (p <= lo + window) {
				candidates[a].push_back (c);
				if (opt.verbose > 1) {
					fprintf (vfd, "Consider min: %f (< %f) chn: %d @ %.2f deg\n", p, lo + window, c, a / (float)kHalfDeg);
				}
			}
		}
	}

	if (stride == 1) { // every angle was measured: the last candidate in ascending order wins (:859-865)
		for (const auto& cand : candidates) {
			for (int c : cand.second) {
				r.angle[c]  = cand.first;
				r.at_min[c] = coarse.chn (c, cand.first);
			}
		}
	} else { // one fine pass of +-(stride+1)/2 half-degrees around every candidate (:866-903)
		const int reach = (stride + 1) / 2;
		for (const auto& cand : candidates) {
			const int centre = cand.first;
			for (prot_engine* g : gpus) {
				gpu_call (prot_engine_reset (g), "reset");
			}
			analysis_pass (gpus, file, n_frames, centre - reach, centre + reach + 1, 1,
			               cand.second.size () > 1 ? -1 : cand.second.front ());
			const Peaks fine (pr, n_chn);
			for (int c : cand.second) {
				for (int a = centre - reach; a < centre + reach + 1; ++a) {
					const float p = fine.chn (opt.link ? -1 : c, a);
					if (p <= r.best[c]) { // ties: the later angle wins
						r.best[c]   = p;
						r.at_min[c] = fine.chn (c, a);
						r.angle[c]  = (a + kAngles) % kAngles;
					}
					if (opt.verbose > 1) {
						print_table_row (fine, n_chn, a);
					}
				}
			}
		}
	}

	// keep the channels' rotations close together: angles beyond 90 degrees may be expressed
	// as negative ones (:905-929)
	float mean = 0;
	int   n_ok = 0;
	for (int c = 0; c < n_chn; ++c) {
		if (r.best[c] != kInf) {
			mean += r.angle[c];
			++n_ok;
		}
	}
	mean /= n_ok;
	const float spread = kAngles / (float)n_ok;
	for (int c = 0; c < n_chn; ++c) {
		if (r.best[c] == kInf) {
			r.angle[c] = 0;
		} else if ((r.angle[c] > 90 * kHalfDeg && fabsf (r.angle[c] - mean) > spread) || mean > 90 * kHalfDeg) {
			r.angle[c] -= kAngles;
		}
	}
	return r;
}

void print_result (const Result& r, int n_chn, FILE* fd) // :931-947
{
	fprintf (fd, "# Result -- Minimize digital peak\n");
	for (int c = 0; c < n_chn; ++c) {
		if (r.best[c] == kInf) {
			fprintf (fd, "Channel: %2d Phase:   0 deg # cannot find min.\n", c + 1);
			continue;
		}
		fprintf (fd, "Channel: %2d Phase: %5.2f deg", c + 1, r.angle[c] / (float)kHalfDeg);
		if (r.angle[c] != 0) {
			fprintf (fd, ", gain: %5.2f dB (att. %4.2f to %4.2f dBFS)", to_dB (r.at_zero[c]) - to_dB (r.at_min[c]),
			         to_dB (r.at_zero[c]), to_dB (r.at_min[c]));
		}
		fprintf (fd, "\n");
	}
}

} // namespace

int main (int argc, char** argv)
{
	Options opt = parse_args (argc, argv);
	// PROT_CLI_TIMING=1: wall time of each phase on stderr (not something the reference prints)
	const bool timing = getenv ("PROT_CLI_TIMING") != nullptr;
	auto       now    = [] () { return std::chrono::steady_clock::now (); };
	auto       t_prev = now ();
	auto       lap    = [&] (const char* what) {
		if (timing) {
			const auto t = now ();
			fprintf (stderr, "[timing] %-28s %9.2f ms\n", what, std::chrono::duration<double, std::milli> (t - t_prev).count ());
			t_prev = t;
		}
	};

	// engines: one, or PROT_GPUS of them on consecutive devices (PROT_GPUS_SAME_DEVICE=1 keeps them
	// all on PROT_DEVICE -- a functional test of the sharding on a single-GPU box).  Created on a second thread as soon as
	// the file's header is known, under the decoding of the samples (the HIP runtime start-up is 0.1-0.2 s).
	const int  n_gpus   = std::max (1, getenv ("PROT_GPUS") ? atoi (getenv ("PROT_GPUS")) : 1);
	const int  dev0     = getenv ("PROT_DEVICE") ? atoi (getenv ("PROT_DEVICE")) : 0;
	const bool same_dev = getenv ("PROT_GPUS_SAME_DEVICE") != nullptr;
	std::vector<prot_engine*> gpus;
	std::string               gpu_err;
	std::thread               gpu_thread;
	auto create_engines = [&] (unsigned blk, int chn) {
		for (int g = 0; g < n_gpus; ++g) {
			prot_config cfg;
			memset (&cfg, 0, sizeof (cfg));
			cfg.kind   = PROT_KIND_CLI;
			cfg.param  = blk;
			cfg.n_chn  = (uint32_t)chn;
			cfg.device = same_dev ? dev0 : dev0 + g;
			int          status = 0;
			prot_engine* e      = prot_engine_create (&cfg, &status);
			if (!e) {
				gpu_err = std::string ("Error: cannot create the GPU engine: ") + prot_strerror (status) + " (" + prot_last_error () + ")\n";
				return;
			}
			gpus.push_back (e);
		}
	};

	wavio::Info        nfo;
	wavio::Samples     file;
	{
		const std::string err = wavio::read_file (opt.in_path, nfo, file, [&] (const wavio::Info& h) {
			const unsigned blk = pick_block_size (opt.blksiz, h.samplerate);
			const int      chn = h.channels;
			gpu_thread         = std::thread ([&create_engines, blk, chn] () { create_engines (blk, chn); });
		});
		if (!err.empty ()) {
			if (gpu_thread.joinable ()) {
				gpu_thread.join ();
			}
			fprintf (stderr, "Cannot open '%s' for reading: ", opt.in_path);
			die (err.c_str ());
		}
	}
	if (gpu_thread.joinable ()) { // (before any of the exits below: no runtime start-up left running behind an exit())
		gpu_thread.join ();
	}
	const int      n_chn    = nfo.channels;
	const uint64_t n_frames = nfo.frames;
	lap ("read + decode input, engine");

	wavio::Writer outfile;
	if (opt.out_path) {
		const std::string err = outfile.open (opt.out_path, nfo);
		if (!err.empty ()) {
			fprintf (stderr, "Cannot open '%s' for writing: ", opt.out_path);
			die (err.c_str ());
		}
	}

	FILE* vfd = opt.verbose > 1 ? stderr : stdout; // :704-706
	if (opt.verbose) {
		fprintf (vfd, "Input File      : %s\n", opt.in_path);
		fprintf (vfd, "Sample Rate     : %d Hz\n", (int)nfo.samplerate);
		fprintf (vfd, "Channels        : %d\n", n_chn);
	}

	std::vector<int> angles;
	const bool       analyse = opt.angle_arg == nullptr;
	if (!analyse) {
		angles = parse_angles (opt.angle_arg, n_chn);
		if (opt.verbose) {
			fprintf (vfd, "# Apply phase-shift\n");
			for (int c = 0; c < n_chn; ++c) {
				fprintf (vfd, "Channel: %2d Phase: %5.2f deg\n", c + 1, angles[c] / (float)kHalfDeg);
			}
		}
	}

	const unsigned blksiz = pick_block_size (opt.blksiz, nfo.samplerate);
	if (opt.verbose > 1) {
		fprintf (vfd, "Process block-size %d\n", blksiz);
	}

	if (!gpu_err.empty () || gpus.empty ()) {
		fputs (gpu_err.empty () ? "Error: cannot create the GPU engine\n" : gpu_err.c_str (), stderr);
		return EXIT_FAILURE;
	}
	prot_engine* pr = gpus[0];

	if (analyse) {
		if (opt.verbose > 1) {
			fprintf (vfd, "Analyzing using %d process threads, stride = %d\n", n_chn, opt.stride);
		}
		if (gpus.size () == 1 && getenv ("PROT_CLI_NO_RESIDENT") == nullptr) {
			// the search re-analyses the whole file once per candidate (cli/phase-rotate.cc:815-929); the samples do not
			// change in between: the engine keeps its de-interleaved copy on the device and only the first pass uploads
			gpu_call (prot_cli_file_resident (pr, 1), "resident file");
		}
		const Result res = find_minimum (gpus, file, n_frames, n_chn, opt, vfd);
		if (gpus.size () == 1 && !opt.out_path) {
			gpu_call (prot_cli_file_resident (pr, 0), "resident file");
		}
		angles           = res.angle;
		if (!opt.out_path || opt.verbose) {
			print_result (res, n_chn, vfd);
		}
		lap ("analyse (find minimum)");
	}

	if (opt.out_path) { // apply and write (:950-1003): latency removed, tail flushed
		if (analyse) {
			gpu_call (prot_engine_reset (pr), "reset");
		}
		std::unique_ptr<float[]> out (new float[(size_t)n_frames * n_chn]); // not zero-filled: every frame is written by the apply
		lap ("allocate output");
		gpu_call (prot_cli_apply_file (pr, file.data (), n_frames, angles.data (), out.get ()), "apply");
		lap ("apply (rotate)");
		if (n_frames != outfile.writef (out.get (), n_frames)) {
			fprintf (stderr, "Error writing to output file.\n");
		}
		outfile.close ();
		lap ("encode + write output");
	}

	for (prot_engine* g : gpus) {
		prot_engine_destroy (g);
	}
	return 0;
}
