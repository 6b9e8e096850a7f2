// ORACLE-SIDE checker glue (test infrastructure, never shipped).
//
// A C entry point over the reference's OWN kaldi-native-fbank sources, compiled
// where they lie under /root/reference/ggml/examples/kaldi-native-fbank/csrc by
// oracle/build_ref.sh into oracle/_ref/libknf_ref.so.  It drives the library
// exactly like the reference's native fbank does
// (ggml/examples/unity/fairseq2.cpp:554-590: NumFrames -> ExtractWindow ->
// FbankComputer::Compute per frame; num_bins=80, 16 kHz, other options default),
// i.e. like fairseq2n's WaveformToFbankConverter that Translator uses
// (src/seamless_communication/inference/translator.py:136-143).
// The waveform is expected already scaled by 2**15.  No standardisation here.
#include <cstdint>
#include <vector>

#include "kaldi-native-fbank/csrc/feature-fbank.h"
#include "kaldi-native-fbank/csrc/feature-window.h"

extern "C" int32_t knf_ref_num_frames(int64_t num_samples) {
    knf::FrameExtractionOptions frame_opts{};
    frame_opts.samp_freq = 16000;
    return knf::NumFrames(num_samples, frame_opts);
}

// out: [num_frames][80]; returns the number of frames written.
extern "C" int32_t knf_ref_fbank(const float* wav, int64_t num_samples, float* out) {
    knf::MelBanksOptions mel_opts{};
    mel_opts.num_bins = 80;
    knf::FrameExtractionOptions frame_opts{};
    frame_opts.samp_freq = 16000;
    frame_opts.dither = 0.0f;
    knf::FbankOptions opts{};
    opts.frame_opts = frame_opts;
    opts.mel_opts = mel_opts;
    const int32_t num_frames = knf::NumFrames(num_samples, frame_opts);
    knf::FbankComputer computer(opts);
    knf::FeatureWindowFunction window_fn(computer.GetFrameOptions());
    std::vector<float> frame;
    for (int32_t f = 0; f < num_frames; ++f) {
        frame.resize(0);
        knf::ExtractWindow(0, wav, (std::size_t)num_samples, f, frame_opts, window_fn, &frame);
        computer.Compute(0.0f, 1.0f, &frame, out + (int64_t)f * 80);
    }
    return num_frames;
}

// The same at another sample rate (window / shift / FFT size / mel banks follow the rate exactly as in the reference's
// FrameExtractionOptions / MelBanksOptions; fairseq2n hands the waveform's own rate through without resampling).
extern "C" int32_t knf_ref_fbank_rate(const float* wav, int64_t num_samples, float sample_rate, float* out) {
    knf::MelBanksOptions mel_opts{};
    mel_opts.num_bins = 80;
    knf::FrameExtractionOptions frame_opts{};
    frame_opts.samp_freq = sample_rate;
    frame_opts.dither = 0.0f;
    knf::FbankOptions opts{};
    opts.frame_opts = frame_opts;
    opts.mel_opts = mel_opts;
    const int32_t num_frames = knf::NumFrames(num_samples, frame_opts);
    knf::FbankComputer computer(opts);
    knf::FeatureWindowFunction window_fn(computer.GetFrameOptions());
    std::vector<float> frame;
    for (int32_t f = 0; f < num_frames; ++f) {
        frame.resize(0);
        knf::ExtractWindow(0, wav, (std::size_t)num_samples, f, frame_opts, window_fn, &frame);
        computer.Compute(0.0f, 1.0f, &frame, out + (int64_t)f * 80);
    }
    return num_frames;
}
