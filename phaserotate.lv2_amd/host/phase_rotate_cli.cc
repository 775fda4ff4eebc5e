// phase_rotate_cli.cc -- `phase-rotate`, the drop-in for cli/phase-rotate.cc on
// MI355X.  Same options, validation, messages, exit codes and result text;
// same analysis procedure (coarse pass -> candidate minima -> fine passes ->
// channel phase-distance unwrapping, cli/phase-rotate.cc:784-947) and apply
// loop (:950-1003).  All DSP -- Hilbert FFT convolution, rotation, the
// per-angle abs-max sweep -- runs in the gfx950 kernels behind the C ABI
// (include/prot_hip.h); this file is option parsing, file I/O and the
// selection logic on the reduced [n_chn][360] peak table.
//
// Differences, all outside the hot path: WAV only (own reader/writer instead
// of libsndfile, no metadata copy); the whole file is read once and kept in
// memory instead of being re-read from disk for every fine pass (:872); a
// partial last block is applied correctly (the reference's tail is broken,
// SURVEY Q3).  No CPU DSP fallback: without a GPU the tool exits with failure.
#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <getopt.h>
#include <limits>
#include <map>
#include <vector>

#include "prot_hip.h"
#include "wavio.h"

#ifndef VERSION
#define VERSION "0.5-mi355x"
#endif

#define SUBSAMPLE PROT_SUBSAMPLE
#define MAXSAMPLE PROT_MAXSAMPLE

static float
coeff_to_dB (float coeff) // cli/phase-rotate.cc:76-83
{
	if (coeff < 1e-15) {
		return -std::numeric_limits<float>::infinity ();
	}
	return 20.0f * log10f (coeff);
}

static void
usage ()
{
	printf ("phase-rotate - Audio File Phase Rotation Util.\n\n");
	printf ("Usage: phase-rotate [ OPTIONS ] <file> [out-file]\n\n");
	printf ("Options:\n"
	        "  -a, --angle <n>[,<n>]*     specify phase angle to apply\n"
	        "  -f, --fftlen <num>         process-block size, freq. resolution\n"
	        "  -h, --help                 display this help and exit\n"
	        "  -l, --link-channels        use downmixed mono peak for analysis\n"
	        "  -s, --stride <num>         analysis step-size\n"
	        "  -v, --verbose              show processing information\n"
	        "  -V, --version              print version information and exit\n"
	        "\n");
	printf ("\n"
	        "This utility analyzes the given audio file to find a phase-rotation\n"
	        "angle that results in minimal digital-peak, while retaining overall\n"
	        "sound and loudness.\n"
	        "\n"
	        "If both input and output file are given, the analysis results applied, and\n"
	        "a new file with optimized phase is written. Otherwise the analysis results\n"
	        "are only printed to standard output.\n"
	        "\n"
	        "Analysis is performed in two steps, first a coarse analysis is performed,\n"
	        "calculating peak for angles distanced `stride' degrees apart. Then local\n"
	        "minimums are explored in a second step.\n"
	        "\n"
	        "Verbose analysis allows to plot the digital peak vs phase-rotation.\n"
	        "The output is in gnuplot(1) data file format.\n"
	        "\n"
	        "If the -a option is specified, no analysis is performed but the given,\n"
	        "phase-angle(s) are directly applied. This requires both input and output\n"
	        "files to be given. If a single angle is given it is applied to all channels\n"
	        "of the file. Otherwise one has to specify the same number of phase-angles as\n"
	        "there are channels in the file.\n"
	        "\n");
	printf ("\n"
	        "Examples:\n"
	        "phase-rotate -l my-music.wav out-file.wav\n\n"
	        "phase-rotate -vv -s 3 my-music.wav\n\n"
	        "phase-rotate -a 10,20 in.wav out.wav\n\n");
	printf ("This build runs the DSP on an AMD MI355X (gfx950) through libprot_hip.\n"
	        "Environment: PROT_DEVICE=<n> selects the HIP device.\n");
	::exit (EXIT_SUCCESS);
}

static void
must (int rc, const char* what)
{
	if (rc != PROT_OK) {
		fprintf (stderr, "Error: %s failed: %s (%s)\n", what, prot_strerror (rc), prot_last_error ());
		::exit (EXIT_FAILURE);
	}
}

// pr.peak(c, a) / peak_all(a) on a host copy of the table (cli/phase-rotate.cc:275-299)
struct PeakTable {
	std::vector<float> v;
	int                n_chn;
	float all (int a) const
	{
		a       = ((a % MAXSAMPLE) + MAXSAMPLE) % MAXSAMPLE;
		float p = 0;
		for (int c = 0; c < n_chn; ++c) {
			p = std::max (p, v[(size_t)c * MAXSAMPLE + a]);
		}
		return p;
	}
	float at (int c, int a) const
	{
		if (c < 0 || c >= n_chn) {
			return all (a);
		}
		a = ((a % MAXSAMPLE) + MAXSAMPLE) % MAXSAMPLE;
		return v[(size_t)c * MAXSAMPLE + a];
	}
};

static PeakTable
fetch_peaks (prot_engine* e, int n_chn)
{
	PeakTable t;
	t.n_chn = n_chn;
	t.v.resize ((size_t)n_chn * MAXSAMPLE);
	must (prot_cli_peaks (e, t.v.data ()), "prot_cli_peaks");
	return t;
}

int
main (int argc, char** argv)
{
	char*        angles_opt = NULL;
	int          stride     = 12 * SUBSAMPLE;
	int          verbose    = 0;
	FILE*        verbose_fd = stdout;
	bool         find_min   = true;
	bool         link_chn   = false;
	unsigned int blksiz     = 0;

	std::vector<int> angles;

	const char*         optstring = "a:f:hls:Vv";
	const struct option longopts[] = {
		{ "angle", required_argument, 0, 'a' },
		{ "fftlen", required_argument, 0, 'f' },
		{ "stride", required_argument, 0, 's' },
		{ "help", no_argument, 0, 'h' },
		{ "link-channels", no_argument, 0, 'l' },
		{ "version", no_argument, 0, 'V' },
		{ "verbose", no_argument, 0, 'v' },
		{ 0, 0, 0, 0 },
	};

	int c = 0;
	while (EOF != (c = getopt_long (argc, argv, optstring, longopts, (int*)0))) {
		switch (c) {
			case 'a': angles_opt = optarg; break;
			case 'f': blksiz = atoi (optarg); break;
			case 'h': usage (); break;
			case 'l': link_chn = true; break;
			case 's': stride = atoi (optarg); break;
			case 'V':
				printf ("phase-rotate version %s\n\n", VERSION);
				printf ("MI355X/HIP implementation of x42 phase-rotate (GPL, Robin Gareus <robin@gareus.org>)\n");
				exit (EXIT_SUCCESS);
				break;
			case 'v': ++verbose; break;
			default:
				fprintf (stderr, "Error: unrecognized option. See --help for usage information.\n");
				::exit (EXIT_FAILURE);
				break;
		}
	}

	if (optind + 1 > argc) {
		fprintf (stderr, "Error: Missing parameter. See --help for usage information.\n");
		::exit (EXIT_FAILURE);
	}
	if (stride < 1 || stride > 45 * SUBSAMPLE || (MAXSAMPLE % stride) != 0) {
		fprintf (stderr, "Error: 180 deg is not evenly dividable by given stride.\n");
		::exit (EXIT_FAILURE);
	}
	if (blksiz != 0 && (blksiz < 1024 || blksiz > 32768)) {
		fprintf (stderr, "Error: fft-len is out of bounds; valid range 1024..32768\n");
		::exit (EXIT_FAILURE);
	}
	if (angles_opt && optind + 2 > argc) {
		fprintf (stderr, "Error: -a, --angle option requires an output file to be given.\n");
		::exit (EXIT_FAILURE);
	}

	wavio::Info        nfo;
	std::vector<float> file;
	{
		std::string err = wavio::read_file (argv[optind], nfo, file);
		if (!err.empty ()) {
			fprintf (stderr, "Cannot open '%s' for reading: ", argv[optind]);
			fputs (err.c_str (), stderr);
			::exit (EXIT_FAILURE);
		}
	}
	const int      n_channels = nfo.channels;
	const uint64_t n_frames   = nfo.frames;

	wavio::Writer outfile;
	const bool    have_out = optind + 1 < argc;
	if (have_out) {
		std::string err = outfile.open (argv[optind + 1], nfo);
		if (!err.empty ()) {
			fprintf (stderr, "Cannot open '%s' for writing: ", argv[optind + 1]);
			fputs (err.c_str (), stderr);
			::exit (EXIT_FAILURE);
		}
	}

	if (verbose > 1) {
		verbose_fd = stderr;
	}
	if (verbose) {
		fprintf (verbose_fd, "Input File      : %s\n", argv[optind]);
		fprintf (verbose_fd, "Sample Rate     : %d Hz\n", (int)nfo.samplerate);
		fprintf (verbose_fd, "Channels        : %d\n", n_channels);
	}

	if (angles_opt) { // cli/phase-rotate.cc:718-747
		find_min = false;
		char* saveptr;
		char* ang;
		while ((ang = strtok_r (angles_opt, ",", &saveptr)) != NULL) {
			char*  ep;
			double a = strtod (ang, &ep);
			if (*ep != '\0' || a < -180 || a > 180) {
				fprintf (stderr, "Error: Invalid angle speficied, value needs to be -180 .. +180.\n");
				::exit (EXIT_FAILURE);
			}
			angles_opt = NULL;
			angles.push_back (round (a * (float)SUBSAMPLE));
		}
		if (angles.size () == 1) {
			while (angles.size () < (size_t)n_channels) {
				angles.push_back (angles.back ());
			}
		}
		if (angles.size () < (size_t)n_channels) {
			fprintf (stderr, "Error: file has more channels than angles were specified.\n");
			::exit (EXIT_FAILURE);
		}
		if (verbose) {
			fprintf (verbose_fd, "# Apply phase-shift\n");
			for (int ch = 0; ch < n_channels; ++ch) {
				fprintf (verbose_fd, "Channel: %2d Phase: %5.2f deg\n", ch + 1, angles[ch] / (float)SUBSAMPLE);
			}
		}
	}

	if (blksiz == 0 || blksiz > 32768) { // :749-755
		blksiz = nfo.samplerate / 8;
	}
	unsigned power_of_two;
	for (power_of_two = 1; 1U << power_of_two < blksiz; ++power_of_two) ;
	blksiz = std::min (32768, std::max (1024, 1 << power_of_two));

	if (verbose > 1) {
		fprintf (verbose_fd, "Process block-size %d\n", blksiz);
	}

	prot_config cfg;
	memset (&cfg, 0, sizeof (cfg));
	cfg.kind   = PROT_KIND_CLI;
	cfg.param  = blksiz;
	cfg.n_chn  = (uint32_t)n_channels;
	cfg.device = getenv ("PROT_DEVICE") ? atoi (getenv ("PROT_DEVICE")) : 0;
	int          status = 0;
	prot_engine* pr     = prot_engine_create (&cfg, &status);
	if (!pr) {
		fprintf (stderr, "Error: cannot create the GPU engine: %s (%s)\n", prot_strerror (status), prot_last_error ());
		::exit (EXIT_FAILURE);
	}

	if (find_min) {
		if (verbose > 1) {
			fprintf (verbose_fd, "Analyzing using %d process threads, stride = %d\n", n_channels, stride);
		}

		// coarse pass over all channels (:784)
		must (prot_cli_analyze_file (pr, file.data (), n_frames, 0, MAXSAMPLE, stride, -1), "analysis");
		PeakTable pk = fetch_peaks (pr, n_channels);

		if (verbose > 1) { // :800-813
			printf ("# Angle mono-peak");
			for (int ch = 0; ch < n_channels; ++ch) {
				printf (" chn-%d", ch + 1);
			}
			printf ("\n");
			for (uint32_t a = 0; a < MAXSAMPLE; a += stride) {
				printf ("%.2f %.4f", a / (float)SUBSAMPLE, coeff_to_dB (pk.all (a)));
				for (int ch = 0; ch < n_channels; ++ch) {
					printf (" %.4f", coeff_to_dB (pk.at (ch, a)));
				}
				printf ("\n");
			}
		}

		// candidate minima per channel (:815-857); key order = ascending angle, as std::map there
		std::map<int, std::vector<int>> mins;
		angles.clear ();
		std::vector<int>   min_angle (n_channels, 0);
		std::vector<float> p_min (n_channels, std::numeric_limits<float>::infinity ());
		std::vector<float> r_zro (n_channels, 0.f), r_min (n_channels, 0.f);

		for (int ch = 0; ch < n_channels; ++ch) {
			float c_min = std::numeric_limits<float>::infinity ();
			float c_max = 0;
			r_zro[ch]   = pk.at (ch, 0);
			for (uint32_t a = 0; a < MAXSAMPLE; a += stride) {
				const float p = pk.at (link_chn ? -1 : ch, a);
				c_min         = std::min (c_min, p);
				c_max         = std::max (c_max, p);
			}
			float range = c_max - c_min;
			if (range == 0) {
				mins[0].push_back (ch); // flat table: "cannot find min." unless refined
				continue;
			}
			if (stride > 1) {
				range *= .07;
				p_min[ch] = std::numeric_limits<float>::infinity ();
			} else {
				range     = 0;
				p_min[ch] = c_min;
			}
			for (uint32_t a = 0; a < MAXSAMPLE; a += stride) {
				const float p = pk.at (link_chn ? -1 : ch, a);
				if (p <= c_min + range) {
					mins[a].push_back (ch);
					if (verbose > 1) {
						fprintf (verbose_fd, "Consider min: %f (< %f) chn: %d @ %.2f deg\n", p, c_min + range, ch, a / (float)SUBSAMPLE);
					}
				}
			}
		}

		if (stride == 1) { // :859-865: the last candidate in ascending order wins
			for (auto& mp : mins) {
				for (auto& cn : mp.second) {
					min_angle[cn] = mp.first;
					r_min[cn]     = pk.at (cn, mp.first);
				}
			}
		} else { // fine passes around every candidate (:866-903)
			const int stride_2 = (stride + 1) / 2;
			for (auto& mp : mins) {
				must (prot_engine_reset (pr), "reset");
				const int ma = mp.first;
				must (prot_cli_analyze_file (pr, file.data (), n_frames, ma - stride_2, ma + stride_2 + 1, 1,
				                             mp.second.size () > 1 ? -1 : mp.second.front ()),
				      "analysis");
				PeakTable fp = fetch_peaks (pr, n_channels);
				for (auto& cn : mp.second) {
					for (int a = ma - stride_2; a < ma + stride_2 + 1; ++a) {
						const float p = fp.at (link_chn ? -1 : cn, a);
						if (p <= p_min[cn]) { // ties: the later angle wins
							p_min[cn]     = p;
							r_min[cn]     = fp.at (cn, a);
							min_angle[cn] = (a + MAXSAMPLE) % MAXSAMPLE;
						}
						if (verbose > 1) {
							printf ("%.2f %.4f", ((a + MAXSAMPLE) % MAXSAMPLE) / (float)SUBSAMPLE, coeff_to_dB (fp.all (a)));
							for (int ch = 0; ch < n_channels; ++ch) {
								printf (" %.4f", coeff_to_dB (fp.at (ch, a)));
							}
							printf ("\n");
						}
					}
				}
			}
		}

		// minimise the phase distance between channels (:905-929)
		float avg_rotate = 0;
		int   avg_count  = 0;
		for (int ch = 0; ch < n_channels; ++ch) {
			if (p_min[ch] != std::numeric_limits<float>::infinity ()) {
				avg_rotate += min_angle[ch];
				++avg_count;
			}
		}
		avg_rotate /= avg_count;
		const float avg_dist = MAXSAMPLE / (float)avg_count;
		for (int ch = 0; ch < n_channels; ++ch) {
			if (p_min[ch] == std::numeric_limits<float>::infinity ()) {
				angles.push_back (0);
			} else {
				if (min_angle[ch] > 90 * SUBSAMPLE && fabsf (min_angle[ch] - avg_rotate) > avg_dist) {
					min_angle[ch] -= MAXSAMPLE;
				} else if (avg_rotate > 90 * SUBSAMPLE) {
					min_angle[ch] -= MAXSAMPLE;
				}
				angles.push_back (min_angle[ch]);
			}
		}

		if (!have_out || verbose) { // :931-947
			fprintf (verbose_fd, "# Result -- Minimize digital peak\n");
			for (int ch = 0; ch < n_channels; ++ch) {
				if (p_min[ch] == std::numeric_limits<float>::infinity ()) {
					fprintf (verbose_fd, "Channel: %2d Phase:   0 deg # cannot find min.\n", ch + 1);
				} else {
					fprintf (verbose_fd, "Channel: %2d Phase: %5.2f deg", ch + 1, min_angle[ch] / (float)SUBSAMPLE);
					if (min_angle[ch] != 0) {
						fprintf (verbose_fd, ", gain: %5.2f dB (att. %4.2f to %4.2f dBFS)",
						         coeff_to_dB (r_zro[ch]) - coeff_to_dB (r_min[ch]), coeff_to_dB (r_zro[ch]), coeff_to_dB (r_min[ch]));
					}
					fprintf (verbose_fd, "\n");
				}
			}
		}
	}

	if (have_out) { // :950-1003
		if (find_min) {
			must (prot_engine_reset (pr), "reset");
		}
		std::vector<float> out ((size_t)n_frames * n_channels);
		must (prot_cli_apply_file (pr, file.data (), n_frames, angles.data (), out.data ()), "apply");
		if (n_frames != outfile.writef (out.data (), n_frames)) {
			fprintf (stderr, "Error writing to output file.\n");
		}
		outfile.close ();
	}

	prot_engine_destroy (pr);
	return 0;
}
