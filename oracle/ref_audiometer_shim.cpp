// TEST INFRASTRUCTURE ONLY.  Drives the REAL reference meter class — AudioMeter::getAudioLevel()
// (/root/reference/audiometer.cpp:16-34), compiled where it lies together with its moc output by
// oracle/Makefile into oracle/_ref/libref_audiometer.so — so the percent scale
// int(float(v*100.0/30000.0)) is pinned to reference object code, not to a restatement.
// A writer thread plays the external VU process: it writes ASCII levels into the FIFO the class
// creates (/tmp/capturefifo<card>), 32-byte NUL-padded records (the class reads 32 bytes at a time).
#include "audiometer.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

extern "C" int ref_audiometer_percent(const char *card, const int *levels, int n, int *percent_out)
{
    const std::string fifo = std::string("/tmp/capturefifo") + card;
    ::unlink(fifo.c_str());
    AudioMeter meter(QString::fromLatin1(card));
    std::vector<int> got;
    QObject::connect(&meter, &AudioMeter::onValueChanged, [&got](int v) { got.push_back(v); });
    std::thread writer([&]() {
        int fd = -1;
        for (int tries = 0; tries < 5000 && fd < 0; ++tries) {      // the class mkfifo()s it itself (audiometer.cpp:20)
            fd = ::open(fifo.c_str(), O_WRONLY | O_NONBLOCK);
            if (fd < 0) ::usleep(1000);
        }
        if (fd < 0) return;
        ::fcntl(fd, F_SETFL, ::fcntl(fd, F_GETFL) & ~O_NONBLOCK);
        for (int i = 0; i < n; ++i) {
            char rec[32];
            std::memset(rec, 0, sizeof rec);
            std::snprintf(rec, sizeof rec, "%d", levels[i]);
            if (::write(fd, rec, sizeof rec) != (ssize_t)sizeof rec) break;
            ::usleep(300);                                          // one record per read(in_fd, in, 32)
        }
        ::close(fd);                                               // read() returns 0 -> getAudioLevel() returns
    });
    meter.getAudioLevel();
    writer.join();
    ::unlink(fifo.c_str());
    const int m = (int)got.size() < n ? (int)got.size() : n;
    for (int i = 0; i < m; ++i) percent_out[i] = got[i];
    return (int)got.size();
}

// Consumer only: run the REAL AudioMeter::getAudioLevel() on /tmp/capturefifo<card> until the (external)
// writer closes the FIFO; returns how many values it emitted.  Used to check that the PRODUCT's FIFO writer
// (igdsp_meter_fifo_*, host library) drives the reference's own meter code.
extern "C" int ref_audiometer_consume(const char *card, int max_out, int *percent_out)
{
    AudioMeter meter(QString::fromLatin1(card));
    std::vector<int> got;
    QObject::connect(&meter, &AudioMeter::onValueChanged, [&got](int v) { got.push_back(v); });
    meter.getAudioLevel();
    const int m = (int)got.size() < max_out ? (int)got.size() : max_out;
    for (int i = 0; i < m; ++i) percent_out[i] = got[i];
    return (int)got.size();
}
