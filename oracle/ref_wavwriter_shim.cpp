// TEST INFRASTRUCTURE ONLY.  extern "C" handle on the REAL reference recorder.
// Compiled together with /root/reference/WavWriter.cpp (where it lies; never
// copied) by oracle/Makefile into oracle/_ref/libref_wavwriter.so.  Drives the
// reference's own WavWriter::start / writeRTPWav / stop (WavWriter.cpp:41-133)
// so tests can compare our recorder output byte-for-byte with the reference's.
#include "WavWriter.h"
#include <string>

extern std::string filePath;   // WavWriter.cpp:195 — directory prefix of generated names
extern FILE *out;              // WavWriter.cpp:5

extern "C" int ref_wav_record(const char *dir_prefix, const char *name_prefix, int rate,
                              const unsigned char *pkt, unsigned pktlen,
                              const unsigned char *payloads, unsigned n_frames, unsigned payloadlen)
{
    WavWriter w;
    filePath = dir_prefix;
    w.start(name_prefix, rate);
    if (!w.isRunning()) return -1;
    for (unsigned f = 0; f < n_frames; ++f)
        w.writeRTPWav((const char *)pkt, (const char *)(payloads + (size_t)f * payloadlen), pktlen, payloadlen);
    w.stop();
    return 0;
}
