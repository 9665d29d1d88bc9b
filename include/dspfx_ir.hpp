// dspfx_ir.hpp -- impulse-response loading for the Fir node, from C++ (the step before the FIR kernel).
//
// Follows nodes/fir.rs:86-173: decode the file to f64 samples, average the channels of every frame
// (`s.iter().sum::<f64>() / num_channels as f64`, fir.rs:140-144), resample to 48 kHz when the file has another rate
// (fir.rs:153-165); the reversal (fir.rs:163,168) happens in dspfx::Fir().  The reference decodes through symphonia and
// resamples through dasp; neither crate is in the reference tree.  This reader handles RIFF/WAVE PCM (8/16/24/32-bit
// integer, 32/64-bit float; integer samples scaled like symphonia's SampleBuffer<f64>: u8 (x-128)/128, i16 /2^15,
// i24 /2^23, i32 /2^31), and `resample_dasp_sinc` restates dasp 0.11.0's Converter + Sinc AS RECALLED -- the same
// restatement, operation for operation, as the Python mirror dsp-stuff_amd/ir.py (tests/test_cpp_ir.py compares them
// bit for bit); only its properties are tested against expectations, the crate's source being absent.
#ifndef DSPFX_IR_HPP
#define DSPFX_IR_HPP

#include <cmath>
#include <cstdint>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace dspfx {

struct IrError : std::runtime_error {
    explicit IrError(const std::string &m) : std::runtime_error("impulse response: " + m) {}
};

struct WavData {
    std::vector<double> samples;   // interleaved [frames][channels]
    unsigned channels = 0, rate = 0;
    std::size_t frames() const { return channels ? samples.size() / channels : 0; }
};

inline WavData read_wav(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw IrError("cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string d = ss.str();
    auto u16 = [&](std::size_t o) { return (unsigned)(unsigned char)d[o] | ((unsigned)(unsigned char)d[o + 1] << 8); };
    auto u32 = [&](std::size_t o) { return (std::uint32_t)u16(o) | ((std::uint32_t)u16(o + 2) << 16); };
    if (d.size() < 12 || d.compare(0, 4, "RIFF") != 0 || d.compare(8, 4, "WAVE") != 0) throw IrError("not a RIFF/WAVE file");
    std::size_t pos = 12, pcm_off = 0, pcm_len = 0;
    bool have_fmt = false, have_data = false;
    unsigned tag = 0, ch = 0, rate = 0, bits = 0;
    while (pos + 8 <= d.size()) {
        const std::string cid = d.substr(pos, 4);
        const std::size_t size = u32(pos + 4), body = pos + 8, avail = body <= d.size() ? std::min(size, d.size() - body) : 0;
        if (cid == "fmt ") {
            if (size < 16 || avail < 16) throw IrError("short fmt chunk");
            tag = u16(body); ch = u16(body + 2); rate = u32(body + 4); bits = u16(body + 14);
            if (tag == 0xFFFE && size >= 26 && avail >= 26) tag = u16(body + 24);   // WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first word
            have_fmt = true;
        } else if (cid == "data") {
            pcm_off = body; pcm_len = avail;
            have_data = true;
        }
        pos += 8 + size + (size & 1);
    }
    if (!have_fmt || !have_data) throw IrError("missing fmt or data chunk");
    if (ch < 1) throw IrError("no channels");
    WavData w;
    w.channels = ch;
    w.rate = rate;
    const unsigned char *p = (const unsigned char *)d.data() + pcm_off;
    std::vector<double> &x = w.samples;
    if (tag == 1) {
        if (bits == 8) for (std::size_t i = 0; i < pcm_len; ++i) x.push_back(((double)p[i] - 128.0) / 128.0);
        else if (bits == 16) for (std::size_t i = 0; i + 2 <= pcm_len; i += 2) x.push_back((double)(std::int16_t)(p[i] | (p[i + 1] << 8)) / 32768.0);
        else if (bits == 24)
            for (std::size_t i = 0; i + 3 <= pcm_len; i += 3) {
                std::int32_t v = (std::int32_t)(p[i] | (p[i + 1] << 8) | (p[i + 2] << 16));
                if (v & 0x800000) v -= 0x1000000;
                x.push_back((double)v / 8388608.0);
            }
        else if (bits == 32)
            for (std::size_t i = 0; i + 4 <= pcm_len; i += 4) {
                const std::uint32_t u = (std::uint32_t)p[i] | ((std::uint32_t)p[i + 1] << 8) | ((std::uint32_t)p[i + 2] << 16) | ((std::uint32_t)p[i + 3] << 24);
                x.push_back((double)(std::int32_t)u / 2147483648.0);
            }
        else throw IrError("unsupported PCM width " + std::to_string(bits));
    } else if (tag == 3) {
        if (bits == 32)
            for (std::size_t i = 0; i + 4 <= pcm_len; i += 4) { float v; std::memcpy(&v, p + i, 4); x.push_back((double)v); }
        else if (bits == 64)
            for (std::size_t i = 0; i + 8 <= pcm_len; i += 8) { double v; std::memcpy(&v, p + i, 8); x.push_back(v); }
        else throw IrError("unsupported float width " + std::to_string(bits));
    } else {
        throw IrError("unsupported WAVE format tag " + std::to_string(tag));
    }
    x.resize(x.size() / ch * ch);
    return w;
}

// fir.rs:153-165: `from_iter(samples).from_hz_to_hz(Sinc::new(Fixed::from([0.0; 16])), rate, 48000).until_exhausted()`,
// dasp 0.11.0 as recalled: a 16-frame ring that new source frames enter at the back, an index that climbs to depth 8, a
// Hann-windowed sinc summed outward from the index with the depth clipped at the ring's ends, the converter stepping
// its interpolation value by source_hz / target_hz and pulling one source frame per whole step.  Sequential f64.
inline std::vector<double> resample_dasp_sinc(const std::vector<double> &src, double source_hz, double target_hz = 48000.0) {
    double ring[16] = {0};
    const int depth = 8;
    int idx = 0;
    const double ratio = source_hz / target_hz, pi = 3.141592653589793;
    double value = 0.0;
    std::size_t pos = 0;
    std::vector<double> out;
    while (true) {
        if (pos >= src.size() && value >= 1.0) break;              // Converter::is_exhausted
        while (value >= 1.0) {                                     // advance whole source frames
            const double nxt = pos < src.size() ? src[pos] : 0.0;  // equilibrium once the iterator has ended
            ++pos;
            for (int k = 0; k < 15; ++k) ring[k] = ring[k + 1];
            ring[15] = nxt;
            if (idx < depth) ++idx;
            value -= 1.0;
        }
        const double phil = value, phir = 1.0 - value;
        const int nl = idx, nr = idx + 1;
        const int rightmost = nl + depth, leftmost = nr - depth;
        const int max_depth = rightmost >= 16 ? 16 - depth : (leftmost < 0 ? depth + leftmost : depth);
        double v = 0.0;
        for (int n = 0; n < max_depth; ++n) {
            double a = pi * (phil + n);
            double first = a == 0.0 ? 1.0 : std::sin(a) / a;
            double second = 0.5 + 0.5 * std::cos(a / depth);
            v = v + (first * second) * ring[((nl - n) % 16 + 16) % 16];
            a = pi * (phir + n);
            first = a == 0.0 ? 1.0 : std::sin(a) / a;
            second = 0.5 + 0.5 * std::cos(a / depth);
            v = v + (first * second) * ring[(nr + n) % 16];        // ring_buffer::Fixed indexes modulo its length
        }
        out.push_back(v);
        value += ratio;
    }
    return out;
}

// h[0..T) in natural order: channels averaged like fir.rs:140-144; another rate than 48 kHz goes through
// resample_dasp_sinc like fir.rs:153-165 (resample = false refuses it instead).  Feed it to dspfx::Fir().
inline std::vector<double> load_impulse_response(const std::string &path, bool resample = true) {
    const WavData w = read_wav(path);
    if (w.rate != 48000 && !resample) throw IrError(std::to_string(w.rate) + " Hz impulse response: convert the file to 48 kHz (or allow resampling)");
    if (w.frames() == 0) throw IrError("empty impulse response");
    std::vector<double> mono(w.frames());
    for (std::size_t i = 0; i < mono.size(); ++i) {
        double acc = 0.0;                                          // Iterator::sum order, then one division
        for (unsigned c = 0; c < w.channels; ++c) acc = acc + w.samples[i * w.channels + c];
        mono[i] = acc / (double)w.channels;
    }
    return w.rate == 48000 ? mono : resample_dasp_sinc(mono, (double)w.rate, 48000.0);
}

}  // namespace dspfx

#endif  // DSPFX_IR_HPP
