// Host-side tables for the fbank kernel: Povey window, mel filter bank, FFT
// factorisation and twiddles.  Computed with the host libm exactly as the
// reference does at session creation (src/fbank.c:49-95,129-171) so the device
// never evaluates pow/cos/log for table entries; twiddles follow pocketfft's
// generator (src/fft/pocketfft.c:65-228,1798-1881) bit for bit.
#pragma once
#include <vector>

namespace aprilx {

struct FbankHostTables {
    int sample_rate = 0, shift = 0, window_size = 0, padded = 0, nfft_bins = 0, nbins = 0;
    std::vector<float> window;            // [padded]
    std::vector<float> mel;               // [nbins][nfft_bins]
    std::vector<int> mel_lo, mel_hi;      // non-zero support per bin
    std::vector<int> factors;             // pocketfft order, e.g. 2,4,4,4,4 for 512; 4,4,5,5 for 400
    std::vector<std::vector<double>> tw;  // per factor (last one empty)
    std::vector<std::vector<double>> tws; // per factor above 5 (generic pass): (cos, sin)(2 pi i / ip), i in [0, ip); else empty
    float pad_value = 0;                  // (float)log((double)kEps)
};

// returns false when the frame length is not supported (an FFT length pocketfft would run through Bluestein's algorithm: a large prime factor)
bool build_fbank_tables(int sample_rate, int frame_shift_ms, int frame_length_ms, int nbins, bool round_pow2,
                        int mel_low, int mel_high, FbankHostTables &out);

}  // namespace aprilx
