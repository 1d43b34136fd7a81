/*
 * basic_pitch_amd.h — C ABI of libbasicpitch_amd.so, the MI355X (gfx950) native executor of the
 * Basic Pitch inference hot path (harmonic CQT + CNN -> note / onset / contour posteriorgrams).
 *
 * This is the drop-in boundary.  The reference (spotify/basic-pitch v0.4.0, pure Python) has no FFI
 * of its own: its plugin point is `basic_pitch.inference.Model` (inference.py:71-182), whose
 * `predict(x)` hands a float32 [n, 43844, 1] batch to a third-party runtime and gets three float32
 * arrays back.  Each entry point below names the reference interface it replaces; INTEGRATION.md
 * shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - every function returns BP_OK (0) or a negative bp_status; the message for the last failure on a
 *     handle is bp_last_error(handle) (bp_last_error(NULL) for bp_create failures);
 *   - outputs are caller-allocated, C-contiguous float32; the library never keeps a caller pointer
 *     after the call returns (weights are copied by bp_create) — the reference's consumer
 *     (note_creation.py:338-341) mutates the returned arrays in place, so they must be caller-owned;
 *   - calls on one handle are serialised by the caller; use one handle per GPU per host thread.
 */
#ifndef BASIC_PITCH_AMD_H
#define BASIC_PITCH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- geometry of the frozen graph (basic_pitch/constants.py:25-47) ---- */
#define BP_AUDIO_SAMPLE_RATE 22050
#define BP_FFT_HOP 256
#define BP_AUDIO_N_SAMPLES 43844 /* constants.py:47  (2 s @ 22.05 kHz minus one hop) */
#define BP_N_FRAMES 172          /* constants.py:44  ANNOT_N_FRAMES */
#define BP_N_BINS_CQT 309        /* models.py:172-177 (103 semitones x 3) */
#define BP_N_FREQ_CONTOUR 264    /* constants.py:36 */
#define BP_N_FREQ_NOTE 88        /* constants.py:35 */
#define BP_N_OVERLAP_FRAMES 30   /* inference.py:190 DEFAULT_OVERLAPPING_FRAMES */
#define BP_OVERLAP_LEN 7680      /* inference.py:304 */
#define BP_HOP_SIZE 36164        /* inference.py:305 */
#define BP_FRAMES_PER_WINDOW 142 /* inference.py:278 (172 - 30) */

typedef struct bp_context* bp_handle;

typedef enum bp_status {
  BP_OK = 0,
  BP_ERR_INVALID_ARG = -1,   /* reference: ValueError (bad shape / null pointer) */
  BP_ERR_BAD_WEIGHTS = -2,   /* reference: ValueError "cannot be loaded" (inference.py:148-154) */
  BP_ERR_NO_DEVICE = -3,     /* no usable gfx950 device / HIP runtime failure at create */
  BP_ERR_HIP = -4,           /* a HIP call failed; see bp_last_error */
  BP_ERR_OUT_OF_MEMORY = -5,
  BP_ERR_UNSUPPORTED = -6,
  BP_ERR_BAD_AUDIO = -7      /* undecodable audio file (reference: librosa.load raises; inference.py:239) */
} bp_status;

/* where `audio` / output pointers live */
typedef enum bp_mem_kind {
  BP_MEM_HOST = 0,  /* pageable or pinned host memory; the library copies H2D / D2H */
  BP_MEM_DEVICE = 1 /* device pointers on the handle's GPU (zero-copy; used by bench.py) */
} bp_mem_kind;

/* bp_create flags */
#define BP_FLAG_STAGE_TIMING 1u /* record HIP events around every stage; read with bp_get_stage_ms */
#define BP_FLAG_F32_MFMA 2u     /* A/B reference path: the whole CNN on the exact-f32 kernels (f32 MFMA
                                   32x32x2 for conv1 layers, f32 VALU heads, 32-channel intermediates in
                                   HBM) instead of the default split-precision path (operands split into
                                   f16 hi + lo, three f16 MFMAs per product, fp32 accumulate: fp32-class
                                   accuracy at the f16 matrix rate; note / onset branches fused) */
#define BP_FLAG_BF16_WEIGHTS 4u /* BASELINE.json configs[3]: the six Conv2D weight tensors are rounded to bf16 at
                                 * bp_create (biases, CQT constants and all activations keep fp32-class precision);
                                 * a bf16 value is one f16 operand, so every conv1 product needs 2 matrix
                                 * instructions instead of 3.  Results follow the graph with bf16-rounded weights to
                                 * the usual 1e-4; against the fp32-weight graph the rounding itself costs up to
                                 * ~5e-3.  Ignored with BP_FLAG_F32_MFMA. */
#define BP_FLAG_EXT_CQT_44K 8u  /* BASELINE.json configs[4]: 44.1 kHz input with an extended CQT range (SURVEY.md App.
                                 * A.6; NOT a behaviour of the reference, which always resamples to 22.05 kHz): the
                                 * same 36 kernels and low-pass at sr = 44100, hop 512, 10 octaves, 345 bins; windows
                                 * are 87,688 samples, track hop 72,328, lead-in 7,680; the CNN and the output shapes
                                 * are unchanged (bins 309..344 feed the harmonic stack where the 22.05 kHz model has
                                 * zeros).  Parity is against the re-parametrised restatement (oracle).  Window /
                                 * track sizes of a handle: bp_handle_window_samples, bp_handle_track_n_*.
                                 * Not available with BP_FLAG_F32_MFMA. */
#define BP_FLAG_TIME_DOMINANT 16u /* record HIP events only around the dominant kernel (stage CONTOUR_CONV1: the folded
                                   * contour conv1, or CONTOUR for BP_CONTOUR_PATH=fused); bp_get_stage_ms then reports that
                                   * stage alone.  Two event records on every FOURTH chunk (the first since creation or
                                   * since the last bp_get_stage_ms included) instead of sixteen on every chunk: the full set costs ~25 us (2.6 %) of a
                                   * 256-window step, and even one pair per chunk 0.04 ms of 0.75 (an event record between
                                   * two kernels keeps the second from starting under the first's tail).
                                   * Ignored with BP_FLAG_STAGE_TIMING / F32_MFMA. */

#define BP_FLAG_F16_CORRECTIONS 32u /* Since round 3 this is the DEFAULT arithmetic and the flag is accepted as a no-op (it
                                    * wins over BP_FLAG_FP8_CORRECTIONS when both are set: the pair is accepted): every matrix product of the path
                                    * is hi hi + lo hi + hi lo on the f16 instruction with fp32 accumulation — fp32-class
                                    * results (<= 2^-22 per product), 5e-6 / 5e-7 per stage against the fp32 oracle. */
#define BP_FLAG_FP8_CORRECTIONS 64u /* NOT IN THE PRODUCT LIBRARY since round 6: bp_create refuses it with BP_ERR_INVALID_ARG
                                    * (unless BP_FLAG_F16_CORRECTIONS is set too, which wins).  Rounds 2 - 5: an opt-in
                                    * reduced-precision mode — the two correction products of the folded contour conv1 and
                                    * of the onset conv1 on the block-scaled fp8 matrix instruction, ~1e-5 / ~3e-5 on the
                                    * contour / onset map; once the default moved to the register-resident marches it was no
                                    * faster (345 k against 347 k windows/s, round 5).  Its kernels are compiled into the A/B
                                    * library only (basic_pitch_amd/build.py: build_library(ab=True)), where the flag works
                                    * as before (tests/test_gpu_parity.py::test_fp8_corrections_mode_lives_in_the_ab_library). */
#define BP_FLAG_BLOCKING_WAIT 128u /* the whole-track calls (bp_infer_track / _tracks / _pcm / _pcm_raw) wait for the device
                                    * asleep on an interrupt instead of spinning on the stream: for file jobs with more worker
                                    * threads than cores (bp_transcribe_files).  Costs tens of microseconds of wake-up
                                    * latency per call; results are the same. */

/*
 * Weights blob ("BPAMDW01", little endian) — produced from the reference's nmp.onnx
 * (basic_pitch/saved_models/icassp_2022/) by basic_pitch_amd/weights.py (at load time when Model() is given
 * the .onnx, ahead of time by tools/extract_weights.py for the shipped assets/nmp_weights.bin):
 *   char magic[8]="BPAMDW01"; u32 version=1; u32 n_tensors;
 *   n_tensors x { char name[24]; u32 ndim; u32 dims[4]; u32 offset_in_floats; u32 count; }
 *   float32 data[]
 * Required tensors: cqt_kernel_re/im [36,256], cqt_lowpass [256], cqt_sqrt_len [309], log_eps [1],
 * log_scale [2], bn_affine [2], {contour1,contour2,note1,note2,onset1,onset2}_{w,b}.
 *
 * Replaces: inference.Model.__init__(model_path) (inference.py:78-154) — load a serialized model.
 * `max_windows_hint` sizes the resident HBM workspace (about 5.8 MB per window); larger batches
 * are processed in chunks of that size.  0 selects the default (256); values above
 * BP_MAX_WINDOWS_PER_CHUNK are rejected with BP_ERR_INVALID_ARG.
 * A handle serves one stream at a time: bp_set_stream orders the new stream after the work already
 * queued on the previous one (the workspace is shared), concurrent calls on one handle are not supported.
 */
#define BP_MAX_WINDOWS_PER_CHUNK 16384
int bp_create(const void* weights, size_t nbytes, int device_ordinal, unsigned flags,
              int64_t max_windows_hint, bp_handle* out);

/* Replaces: Model going out of scope. */
void bp_destroy(bp_handle h);

/* Error text for the last failing call on `h` (NULL: last bp_create failure in this thread). */
const char* bp_last_error(bp_handle h);

/*
 * Replaces: Model.predict(x) (inference.py:156-182) for x = float32 [n_windows, 43844(,1)].
 *   note    float32 [n_windows, 172,  88]   (ONNX output StatefulPartitionedCall:1)
 *   onset   float32 [n_windows, 172,  88]   (StatefulPartitionedCall:2)
 *   contour float32 [n_windows, 172, 264]   (StatefulPartitionedCall:0)
 * Blocking: returns after the outputs are complete (host) or enqueued+synchronised (device).
 */
int bp_infer(bp_handle h, const float* audio, int64_t n_windows, float* note, float* onset,
             float* contour, int mem_kind);

/* Same as bp_infer with BP_MEM_DEVICE but only enqueues on the handle's stream (no sync). */
int bp_infer_async(bp_handle h, const float* audio_dev, int64_t n_windows, float* note_dev,
                   float* onset_dev, float* contour_dev);

/*
 * Replaces: run_inference() minus decode (inference.py:282-315): get_audio_input's 3840-sample zero
 * lead-in + window_audio_file (inference.py:194-244), the per-window predict loop (308-310) and
 * unwrap_output (247-279), all on the device.  `samples` = mono 22.05 kHz float32 [n_samples].
 * Outputs have bp_track_n_frames(n_samples) rows: note/onset [T,88], contour [T,264].
 */
int bp_infer_track(bp_handle h, const float* samples, int64_t n_samples, float* note, float* onset,
                   float* contour, int mem_kind);

/*
 * Many tracks in one call: the per-file loop of predict_and_save (inference.py:548-604) with the windows of
 * consecutive tracks packed into full batches (a 3-minute track is only 110 windows; one track per launch leaves a
 * third of the GPU idle).  samples[i] = mono 22.05 kHz float32 [n_samples[i]]; note[i] / onset[i] / contour[i] as
 * bp_infer_track for track i.  Results are bit-identical to n_tracks separate bp_infer_track calls.
 */
int bp_infer_tracks(bp_handle h, int64_t n_tracks, const float* const* samples, const int64_t* n_samples,
                    float* const* note, float* const* onset, float* const* contour, int mem_kind);

/*
 * Replaces: the decode-side tail of librosa.load(path, sr=22050, mono=True) (inference.py:239) for PCM that is
 * already decoded: channel-mean downmix (librosa.to_mono) + rational polyphase resampling to 22.05 kHz, on the
 * device.  `pcm` = interleaved float32 [n_frames][channels] at `sample_rate` Hz (host or device per mem_kind).
 * The resampler has the response of librosa's default res_type "soxr_hq" (libsoxr's SOXR_HQ design restated: linear
 * phase, pass-band to 0.9136 of the lower Nyquist, 126 dB from that Nyquist on; one zero-phase polyphase stage with
 * float64 taps and accumulation — audio_ingest.hip); with it the reference's golden posteriorgrams for
 * its 44.1 kHz clip are met at its own atol = 1e-4.
 *   bp_resampled_length  ceil(n_frames * 22050 / sample_rate) samples (librosa.resample's output length)
 *   bp_resample          the mono 22.05 kHz signal itself -> out22k [bp_resampled_length] (host or device)
 *   bp_infer_pcm         bp_resample + bp_infer_track without the signal leaving the device; outputs as
 *                        bp_infer_track with T = bp_track_n_frames(bp_resampled_length(n_frames, sample_rate))
 * Errors: BP_ERR_INVALID_ARG for channels < 1 or > 64, sample_rate outside [1000, 768000].  Any rate ratio is
 * accepted (irregular ones evaluate the filter taps in the kernel instead of from a table).
 */
int64_t bp_resampled_length(int64_t n_frames, int sample_rate);
int bp_resample(bp_handle h, const float* pcm, int64_t n_frames, int channels, int sample_rate, float* out22k,
                int mem_kind);
int bp_infer_pcm(bp_handle h, const float* pcm, int64_t n_frames, int channels, int sample_rate, float* note,
                 float* onset, float* contour, int mem_kind);

/*
 * bp_infer_pcm on the samples as the file stores them: interleaved little-endian PCM in one of the formats below goes over
 * PCIe as it is (16-bit stereo: half the bytes of its float form) and becomes float on the device, with the scaling of
 * libsndfile's float read that librosa.load uses (inference.py:239): integers / 2^(bits-1), 8-bit unsigned (v - 128) / 128,
 * float64 rounded to float32.  Same result, bit for bit, as converting on the host and calling bp_infer_pcm.
 * bp_host_alloc / bp_host_free: page-locked host memory (any thread, any device); copies to and from it are DMA without
 * the runtime's staging copy — for `pcm` and the three outputs of the BP_MEM_HOST calls.  NULL when the allocation fails.
 */
typedef enum bp_pcm_format {
  BP_PCM_F32 = 0, /* IEEE float32 */
  BP_PCM_S16 = 1,
  BP_PCM_S24 = 2, /* packed, 3 bytes per sample */
  BP_PCM_S32 = 3,
  BP_PCM_U8 = 4,
  BP_PCM_F64 = 5
} bp_pcm_format;
int bp_infer_pcm_raw(bp_handle h, const void* pcm, int format, int64_t n_frames, int channels, int sample_rate, float* note,
                     float* onset, float* contour, int mem_kind);
void* bp_host_alloc(size_t bytes);
void bp_host_free(void* p);

/*
 * Replaces: the decode step of librosa.load for FLAC input (inference.py:239; README.md:182-189 lists .flac) — host
 * code, no handle, thread-safe.  `file` = the whole file in memory.
 *   bp_flac_info    channels, sample rate, bits per sample and frame count from STREAMINFO (frames are counted by
 *                   decoding when STREAMINFO leaves the count at 0)
 *   bp_flac_decode  interleaved float32 [n_frames][channels] in [-1, 1) (value / 2^(bits-1), libsndfile's float
 *                   conversion) into pcm[capacity_frames][channels]; verifies the CRC-8 / CRC-16 of every frame
 *                   and the MD5 of the decoded samples against STREAMINFO
 * Errors: BP_ERR_BAD_AUDIO (message from bp_audio_last_error, thread-local), BP_ERR_INVALID_ARG.
 */
int bp_flac_info(const void* file, size_t nbytes, int* channels, int* sample_rate, int* bits_per_sample,
                 int64_t* n_frames);
int bp_flac_decode(const void* file, size_t nbytes, float* pcm, int64_t capacity_frames, int64_t* n_frames);

/* FLAC decode ON THE DEVICE (round 6; csrc/flac_device.hip): the file's bytes go over PCIe as they are (about half the PCM's)
 * and three launches decode them — a scan for frame headers (sync code + CRC-8 + STREAMINFO's sample size / channel count), a
 * one-lane walk that keeps the chain of consecutive frame / sample numbers, and one lane per frame for the serial part (Rice
 * residuals, prediction, stereo decorrelation, CRC-16) — into the interleaved 16- / 32-bit PCM the ingest kernels read.  The
 * samples are bit-identical to bp_flac_decode's.  What it leaves to the host decoder (BP_ERR_UNSUPPORTED): streams whose
 * STREAMINFO lacks the sample count or the block sizes, more than 24 bits per sample, more than 8 channels; a stream the
 * device cannot follow (lost chain, CRC-16 mismatch, reserved values) is BP_ERR_BAD_AUDIO — the host decoder then says
 * why.  The MD5 of STREAMINFO is not checked on the device; every frame's CRC-16 and the sample count are.
 *   bp_flac_layout          STREAMINFO without decoding (host): what the device call needs to size its buffers
 *   bp_flac_decode_device   the samples back on the host as interleaved int32 (sign-extended): the test / tool entry
 *   bp_infer_flac           the posteriorgrams of a FLAC file's bytes: device decode + what bp_infer_pcm_raw does
 *   bp_infer_flac_candidates  ... + the device half of note decoding (bp_infer_pcm_raw_candidates) */
typedef struct bp_flac_stream_layout {
  int32_t channels, sample_rate, bits_per_sample, min_block, max_block;
  int64_t n_frames;     /* samples per channel; 0: unknown */
  int64_t audio_start;  /* byte offset of the first frame */
} bp_flac_stream_layout;
int bp_flac_layout(const void* file, size_t nbytes, bp_flac_stream_layout* out);
int bp_flac_decode_device(bp_handle h, const void* file, size_t nbytes, int32_t* pcm, int64_t capacity_frames, int64_t* n_frames);
int bp_infer_flac(bp_handle h, const void* file, size_t nbytes, float* note, float* onset, float* contour, int mem_kind);
const char* bp_audio_last_error(void);

/* ceil((n_samples + 3840) / 36164) windows (inference.py:207,242); 0 for n_samples <= 0 */
int64_t bp_track_n_windows(int64_t n_samples);
/* min(n_windows*142, int(n_samples / 36164 * 142)) rows (inference.py:277-279) */
int64_t bp_track_n_frames(int64_t n_samples);

/* The same counts for a handle's geometry (identical to the functions above unless BP_FLAG_EXT_CQT_44K), the samples
 * per window bp_infer expects (43844 / 87688), the sample rate bp_infer_track expects (22050 / 44100) and the length
 * bp_resample / bp_infer_pcm produce for n_frames at sample_rate. */
int64_t bp_handle_track_n_windows(bp_handle h, int64_t n_samples);
int64_t bp_handle_track_n_frames(bp_handle h, int64_t n_samples);
int64_t bp_handle_window_samples(bp_handle h);
int bp_handle_sample_rate(bp_handle h);
int64_t bp_handle_resampled_length(bp_handle h, int64_t n_frames, int sample_rate);

/* Run on an externally owned hipStream_t (e.g. torch's current stream); NULL = library stream. */
int bp_set_stream(bp_handle h, void* hip_stream);
int bp_synchronize(bp_handle h);

/* ---- introspection ---- */
typedef struct bp_info {
  int device_ordinal;
  int compute_units;
  int64_t max_windows;     /* resident workspace capacity (windows per chunk) */
  int64_t workspace_bytes; /* HBM held by this handle */
  char arch[32];           /* "gfx950..." */
} bp_info;
int bp_get_info(bp_handle h, bp_info* out);

/* Stage ids for bp_get_stage_ms / bp_run_stage. */
enum {
  BP_STAGE_PYRAMID = 0,   /* 8 x decimate-by-2          audio -> pyr          */
  BP_STAGE_FILTERBANK = 1,/* 9-level CQT filterbank     audio,pyr -> lp, mm   */
  BP_STAGE_CONTOUR1 = 2,  /* norm+BN+stack+conv 3x39    lp,mm -> c1           */
  BP_STAGE_CONTOUR2 = 3,  /* conv 5x5 + sigmoid         c1 -> contour         */
  BP_STAGE_NOTE1 = 4,     /* conv 7x7 s3 + ReLU         contour -> n1         */
  BP_STAGE_NOTE2 = 5,     /* conv 7x3 + sigmoid         n1 -> note            */
  BP_STAGE_ONSET1 = 6,    /* norm+BN+stack+conv 5x5 s3  lp,mm -> o1           */
  BP_STAGE_ONSET2 = 7,    /* concat + conv 3x3 + sigm.  note,o1 -> onset      */
  BP_STAGE_ZPACK = 8,     /* norm + BN, f16 hi|lo words  lp,mm -> zp          */
  BP_STAGE_NOTE = 9,      /* fused note branch           contour -> note      */
  BP_STAGE_ONSET = 10,    /* fused onset branch          zp,note -> onset     */
  BP_STAGE_CONTOUR = 11,  /* contour branch (bp_run_stage: both kernels; timing: the fused A/B kernel) */
  BP_STAGE_CONTOUR_CONV1 = 12, /* contour conv 3x39 + ReLU on the matrix cores   zp -> c1 (internal)   */
  BP_STAGE_CONTOUR_CONV2 = 13, /* contour conv 5x5 + sigmoid                      c1 -> contour         */
  BP_STAGE_CONTOUR_CONV1_EDGE = 14, /* the rim groups of CONTOUR_CONV1 on the exact 8-channel kernel (CONTOUR_CONV1 is
                                     * then the folded kernel over the interior groups)                              */
  BP_N_STAGES = 15
};
/* Default (split-precision) path runs: PYRAMID, FILTERBANK, ZPACK, CONTOUR_CONV1_EDGE, CONTOUR_CONV1, CONTOUR_CONV2, NOTE, ONSET.
 * BP_FLAG_F32_MFMA runs:               PYRAMID, FILTERBANK, CONTOUR1, CONTOUR2, NOTE1, NOTE2, ONSET1, ONSET2. */

/* With BP_FLAG_STAGE_TIMING (or BP_FLAG_TIME_DOMINANT, one stage): mean milliseconds per stage over the chunks (<= 128 most recent) run
 * since the previous call (0 for stages the handle's path does not run); n >= BP_N_STAGES;
 * synchronises the stream and resets the accumulation. */
int bp_get_stage_ms(bp_handle h, float* ms, int n);

/*
 * Test hook: run ONE stage on caller-supplied DEVICE buffers (layouts in DESIGN.md "HBM layout"):
 *   audio [n,43844]  pyr [n,BP_PYR_STRIDE]  lp [n,172,309]  mm int32 [n,2] (ordered-int min,max)
 *   c1 [n,8,172,264]  contour [n,172,264]  n1 [n,32,172,88]  note [n,172,88]  o1 [n,32,172,88]
 *   onset [n,172,88]  zp uint32 [n,BP_Z_ROWS,BP_Z_ROW] (z = normalised, BatchNorm-ed CQT as f16 hi | f16 lo << 16
 *   with lo = (z - hi) * 2^11; frame t, bin g at [t + 1][BP_Z_PAD + g], everything else zero: the zero padding of
 *   the harmonic stack and of the frame halo is part of the tensor).  Unused pointers for a stage may be NULL.
 *   Synchronous.
 */
#define BP_Z_ROW 448
#define BP_Z_ROWS 174
#define BP_Z_PAD 56
#define BP_PYR_STRIDE 43712
typedef struct bp_stage_buffers {
  const float* audio;
  float* pyr;
  float* lp;
  int32_t* mm;
  float* c1;
  float* contour;
  float* n1;
  float* note;
  float* o1;
  float* onset;
  uint32_t* zp;
} bp_stage_buffers;
int bp_run_stage(bp_handle h, int stage, const bp_stage_buffers* buf, int64_t n_windows);

/* Offsets (in floats) of pyramid levels 1..8 inside one window's pyr row, and their lengths. */
int bp_pyramid_layout(int level /*1..8*/, int64_t* offset, int64_t* length);

const char* bp_version(void);
/* HIP devices visible to this process (0 without a GPU). */
int bp_device_count(void);

/* ---- note decoding: posteriorgrams -> note events (host C++, no GPU needed) -------------------------
 * Replaces the Python loops of basic_pitch/note_creation.py: output_to_notes_polyphonic (360-511, with
 * constrain_frequency 314-343, get_infered_onsets 289-311 and the melodia trick 452-509), get_pitch_bends
 * (182-219) and model_frames_to_time (346-357), i.e. model_output_to_notes (52-116) without the PrettyMIDI
 * object.  Events come back in the reference's order with the reference's values (bit-exact vs the numpy
 * restatement in oracle/note_oracle.py, which reproduces the reference's golden events). */
typedef struct bp_note_params {
  double onset_threshold;      /* predict(onset_threshold=0.5)                          inference.py:434 */
  double frame_threshold;      /* predict(frame_threshold=0.3)                          inference.py:435 */
  double min_freq_hz;          /* <= 0: None                                            inference.py:437 */
  double max_freq_hz;          /* <= 0: None                                            inference.py:438 */
  int32_t min_note_len;        /* frames: int(round(ms / 1000 * (22050 / 256)))         inference.py:469 */
  int32_t infer_onsets;        /* 1                                                     note_creation.py:56 */
  int32_t melodia_trick;       /* 1                                                     inference.py:440 */
  int32_t include_pitch_bends; /* 1                                                     note_creation.py:60 */
  int32_t energy_tol;          /* 11                                                    note_creation.py:370 */
  int32_t reserved;
} bp_note_params;

typedef struct bp_note_event {
  double start_s, end_s;       /* model_frames_to_time()[start_frame / end_frame] */
  int64_t bend_offset;         /* first pitch bend of this note in `bends` */
  int32_t start_frame, end_frame;
  int32_t pitch_midi;          /* 21 + note bin */
  int32_t n_bends;             /* end_frame - start_frame, or 0 without pitch bends */
  float amplitude;             /* np.mean(frames[start:end, bin]) in float32 */
  int32_t reserved;
} bp_note_event;

void bp_note_params_default(bp_note_params* p);

/* note / onset [n_frames, 88] are MODIFIED in place when min/max frequency is set, exactly like the
 * reference's constrain_frequency (note_creation.py:338-341); contour [n_frames, 264] is read only.
 * On return *n_events / *n_bends hold the required counts; if they exceed max_events / max_bends the call
 * fails with BP_ERR_INVALID_ARG and can be repeated with larger buffers.  bends are in 1/3 semitones. */
int bp_notes_decode(float* note, float* onset, const float* contour, int64_t n_frames,
                    const bp_note_params* params, bp_note_event* events, int64_t max_events, int32_t* bends,
                    int64_t max_bends, int64_t* n_events, int64_t* n_bends);

/* ---- Note decoding with the dense half on the device (round 5; SURVEY.md 8f rank 1: onset inference, peak picking and
 * thresholding note_creation.py:289-311, 394-402; the pitch bends of get_pitch_bends 182-219 for every (frame, bin)).
 * What the sequential note tracker needs of a track's posteriorgrams is the note map, WHERE the onset peaks are and the
 * pitch bend per (frame, note bin): 7.0 MB per 3-minute track instead of 27.6 MB across PCIe, and the three dense scans
 * leave the host cores.  Same events as bp_notes_decode, bit for bit (tests/test_gpu_parity.py).
 *
 *   note_out  [n_frames][88] float32  the note map, frequency-constrained like constrain_frequency (314-343)
 *   cand_bits [n_frames][12] bytes    bit (f & 7) of byte f >> 3 in row t: (t, f) is an onset peak >= onset_threshold
 *                                     (BP_NOTE_CAND_ROW_BYTES: 88 bits and a zero byte — rows of three 32-bit words)
 *   bend_map  [n_frames][88] int8     pitch bend of note bin f at frame t in 1/3 semitones; may be NULL without pitch bends
 *   *status   0: decode with bp_notes_decode_candidates;  1: the maps hold a NaN or onset_threshold <= 0 (numpy's
 *             propagation rules / every non-peak qualifies): decode the maps themselves with bp_notes_decode.
 * bp_note_candidates takes the three maps from host or device memory (mem_kind) and leaves them untouched;
 * bp_infer_pcm_raw_candidates is bp_infer_pcm_raw (inference.py:239 onwards) whose posteriorgrams stay on the device.
 * All outputs are host buffers; into page-locked ones (bp_host_alloc) a kernel writes them across PCIe itself, so that the
 * copy engine — which serialises the copies of all handles, in both directions — is left to the inbound samples. */
#define BP_NOTE_CAND_ROW_BYTES 12
int bp_note_candidates(bp_handle h, const float* note, const float* onset, const float* contour, int64_t n_frames,
                       const bp_note_params* params, int mem_kind, float* note_out, uint8_t* cand_bits, int8_t* bend_map,
                       int* status);
int bp_infer_pcm_raw_candidates(bp_handle h, const void* pcm, int format, int64_t n_frames, int channels, int sample_rate,
                                const bp_note_params* params, float* note_out, uint8_t* cand_bits, int8_t* bend_map,
                                int* status);
/* the same from a FLAC file's bytes, decoded on the device (bp_infer_flac) */
int bp_infer_flac_candidates(bp_handle h, const void* file, size_t nbytes, const bp_note_params* params, float* note_out,
                             uint8_t* cand_bits, int8_t* bend_map, int* status);
/* The three posteriorgrams the LAST bp_infer_*_candidates call of this handle left on the device, n_frames rows each
 * ([n][88], [n][88], [n][264]; host or device destinations): what that call's *status == 1 asks for (a NaN in the maps: the
 * host decoder needs the maps themselves, bp_notes_decode) without running the track again.  BP_ERR_INVALID_ARG when
 * n_frames is not the row count of that call or another call has used the handle's track buffer since. */
int bp_track_maps(bp_handle h, int64_t n_frames, float* note, float* onset, float* contour, int mem_kind);
/* The sequential half (host): output_to_notes_polyphonic from the candidates (note_creation.py:404-509), pitch bends read
 * from bend_map, frame times as bp_notes_decode.  `note` is only read. */
int bp_notes_decode_candidates(const float* note, const uint8_t* cand_bits, const int8_t* bend_map, int64_t n_frames,
                               const bp_note_params* params, bp_note_event* events, int64_t max_events, int32_t* bends,
                               int64_t max_bends, int64_t* n_events, int64_t* n_bends);
const char* bp_notes_last_error(void);

/* ---- whole files, natively: decode -> posteriorgrams -> note events -> .mid / .csv (host C++ threads) ----------
 * Replaces the per-file Python loop of predict_and_save (basic_pitch/inference.py:509-604) for WAV and FLAC input:
 * librosa.load (239: decode; downmix + resampling on the device), run_inference (282-330), model_output_to_notes
 * (note_creation.py:52-116), note_events_to_midi + pretty_midi write (222-267; inference.py:586) and save_note_events
 * (inference.py:409-428).  A worker thread owns a file from its bytes to its outputs; the handles are GPU lanes the
 * workers queue for (three handles per GPU overlap one file's transfers with another's kernels; handles of several GPUs
 * in one call shard the files over them: a worker takes whichever lane is free, all handles of the same mode).  Outputs are named
 * <out_dir>/<stem>_basic_pitch.mid / .csv and never overwritten (inference.py:401-404): an existing file, and every
 * later input with the stem of an earlier one, is reported instead of written.  Per-file failures do not stop the job
 * (the reference's per-file try / except, 548-604): reports[i].status / message.  Bytes are identical to what
 * basic_pitch_amd.predict_and_save writes (tests/test_file_pipeline.py). */
typedef struct bp_transcribe_params {
  bp_note_params notes;         /* bp_note_params_default */
  double midi_tempo;            /* predict(midi_tempo=120)                               inference.py:443 */
  int32_t multiple_pitch_bends; /* 0                                                     inference.py:439 */
  int32_t save_midi;            /* 1 */
  int32_t save_notes;           /* 1 */
  int32_t threads;              /* host worker threads; <= 0: one per hardware thread (at most 64) */
  int32_t host_decode;          /* 0 (default): the dense half of note decoding runs on the device and the note map, the
                                   onset-peak bitmap and the pitch-bend map come back (bp_infer_pcm_raw_candidates: 7 MB per
                                   3-minute track); 1: all three posteriorgrams come back (27.6 MB) and the host decodes
                                   them (bp_notes_decode) — the round-4 path, same events */
  int32_t direct_io;            /* 0 (default): files are read through the page cache; 1: with O_DIRECT, the storage device's
                                   DMA landing in the page-locked buffer the GPU copies from (no page-cache copy: half the
                                   host-DRAM traffic per file and no core-ms for it; a file system that refuses O_DIRECT is
                                   read buffered).  For corpora that do NOT fit the page cache — a file that is already
                                   cached is read faster from there */
  int32_t host_flac;            /* 0 (default): FLAC files are decoded on the device (bp_infer_flac: the file's bytes over PCIe,
                                   no host core-time per sample; streams it leaves to the host fall back by themselves);
                                   1: on the host (bp_flac_decode), the round-4 path */
  int32_t reserved[1];
} bp_transcribe_params;

typedef struct bp_file_report {
  int32_t status;               /* BP_OK or the bp_status of the step that failed */
  int32_t n_note_events;
  int64_t n_frames;             /* rows of the file's posteriorgrams */
  char message[240];            /* empty on success */
  /* where the worker's wall time for this file went, in milliseconds: reading the file (and FLAC decode), waiting for a
   * GPU lane, the device call (copy in, resample, CQT + CNN, copy out), note decoding, MIDI / CSV encoding + writes */
  float ms_read, ms_lane_wait, ms_device, ms_notes, ms_write;
} bp_file_report;

void bp_transcribe_params_default(bp_transcribe_params* p);
int bp_transcribe_files(bp_handle* handles, int n_handles, const char* const* paths, int64_t n_files, const char* out_dir,
                        const bp_transcribe_params* params, bp_file_report* reports);
/* The pieces on their own (host only; the CPU tests pin them to the Python writers and readers).
 * bp_notes_to_midi / bp_notes_to_csv return the number of bytes of the file (written to `out` if it fits `capacity`; call
 * with out = NULL to size the buffer), or a negative bp_status.  `bends` may be NULL (no pitch bends).
 * bp_wav_info / bp_wav_decode: RIFF/WAVE PCM 8 / 16 / 24 / 32 and IEEE float 32 / 64 -> float32 [n_frames, channels] in
 * [-1, 1), the scaling of libsndfile's float read that librosa.load uses (inference.py:239). */
int64_t bp_notes_to_midi(const bp_note_event* events, int64_t n_events, const int32_t* bends, int multiple_pitch_bends,
                         double midi_tempo, uint8_t* out, int64_t capacity);
int64_t bp_notes_to_csv(const bp_note_event* events, int64_t n_events, const int32_t* bends, char* out, int64_t capacity);
int bp_wav_info(const void* file, size_t nbytes, int* channels, int* sample_rate, int* bits_per_sample, int64_t* n_frames);
int bp_wav_decode(const void* file, size_t nbytes, float* pcm, int64_t max_frames, int64_t* n_frames);
const char* bp_files_last_error(void); /* thread-local */
/* bp_transcribe_files keeps its workers' page-locked buffers (a file's bytes, its posteriorgrams) in a process-wide pool
 * between calls; this releases the pooled ones (a long-lived service calls it when a burst of jobs is over). */
void bp_files_release_buffers(void);
/* files whose bytes bp_transcribe_files read with O_DIRECT since the library was loaded (params.direct_io; 0 when the file
 * system refused the flag) */
int64_t bp_files_direct_reads(void);
/* bp_transcribe_files' file reader on its own, into ordinary host memory (test hook, no device): returns the number of bytes
 * read (or a negative bp_status), a 64-bit FNV-1a of them and whether O_DIRECT was really used. */
int64_t bp_files_read_probe(const char* path, int direct_io, uint64_t* fnv1a, int* used_direct);

#ifdef __cplusplus
}
#endif
#endif /* BASIC_PITCH_AMD_H */
