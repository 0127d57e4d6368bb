"""
Live-stream noise suppression: the reference's microphone demo (demo/python/koala_demo_mic.py:75-121: a recorder hands
over one frame at a time, every frame goes through `Koala.process`, the enhanced -- and optionally the original --
audio is appended to a WAV file until Ctrl+C) restated on koala_amd.  This environment has no capture device and no
`pvrecorder`, so the frame source is a byte stream of raw 16 kHz mono int16 PCM: stdin by default (e.g.
`arecord -f S16_LE -r 16000 -c 1 -t raw | python -m koala_amd.demo.koala_demo_stream --output_path out.wav`), or a WAV
file replayed with `--realtime` pacing.  One frame per call = the hipGraph latency path (BASELINE configs[4]).

    python -m koala_amd.demo.koala_demo_stream --output_path clean.wav < noisy.raw
    python -m koala_amd.demo.koala_demo_stream --input_path noisy.wav --realtime --output_path clean.wav --reference_output_path in.wav
"""
import argparse
import contextlib
import struct
import sys
import time
import wave

import numpy as np

import koala_amd


def frames_from_raw(stream, frame_length):
    """Yields int16 frames from a binary stream; a trailing partial frame is zero-padded (the recorder never makes one)."""
    nbytes = 2 * frame_length
    while True:
        buf = stream.read(nbytes)
        if not buf:
            return
        while len(buf) < nbytes:  # pipes deliver short reads
            more = stream.read(nbytes - len(buf))
            if not more:
                break
            buf += more
        frame = np.zeros(frame_length, np.int16)
        got = np.frombuffer(buf[:len(buf) // 2 * 2], dtype='<i2')
        frame[:len(got)] = got
        yield frame
        if len(buf) < nbytes:
            return


def frames_from_wav(path, frame_length, sample_rate):
    with wave.open(path, 'rb') as f:
        if f.getframerate() != sample_rate or f.getnchannels() != 1 or f.getsampwidth() != 2:
            raise ValueError('`%s` must be %d Hz, single-channel, 16-bit PCM' % (path, sample_rate))
        pcm = np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16)
    for start in range(0, len(pcm), frame_length):
        frame = np.zeros(frame_length, np.int16)
        chunk = pcm[start:start + frame_length]
        frame[:len(chunk)] = chunk
        yield frame


def main(argv=None):
    ap = argparse.ArgumentParser(description='frame-by-frame noise suppression of a live PCM stream')
    ap.add_argument('--access_key', default='koala-amd', help='accepted for interface compatibility; not checked')
    ap.add_argument('--input_path', default=None, help='WAV file to replay instead of raw PCM on stdin')
    ap.add_argument('--output_path', default=None, help='WAV file for the enhanced audio')
    ap.add_argument('--reference_output_path', default=None, help='WAV file for the unprocessed input')
    ap.add_argument('--model_path', default=None)
    ap.add_argument('--library_path', default=None)
    ap.add_argument('--device', default=None)
    ap.add_argument('--precision', default=None, choices=['fp32', 'bf16'],
                    help='mask network arithmetic (default: the library default, fp32, or $KOALA_AMD_PRECISION)')
    ap.add_argument('--realtime', action='store_true', help='pace a replayed file at 16 ms per frame, like a recorder')
    ap.add_argument('--show_devices', action='store_true')
    args = ap.parse_args(argv)

    if args.show_devices:
        print('\n'.join(koala_amd.available_devices(library_path=args.library_path)))
        return 0
    if args.output_path is None:
        raise ValueError('Missing required argument --output_path')
    for p in (args.output_path, args.reference_output_path):
        if p is not None and not p.lower().endswith('.wav'):
            raise ValueError('Given output paths must have WAV file extension')

    if args.precision is not None:
        import os
        os.environ['KOALA_AMD_PRECISION'] = args.precision
    koala = koala_amd.create(access_key=args.access_key, model_path=args.model_path, device=args.device,
                             library_path=args.library_path)
    print('Koala version: %s' % koala.version)
    n = koala.frame_length
    source = (frames_from_wav(args.input_path, n, koala.sample_rate) if args.input_path
              else frames_from_raw(sys.stdin.buffer, n))
    lat = []
    frames = 0
    t_start = time.perf_counter()
    try:
        with contextlib.ExitStack() as stack:
            def open_wav(path):
                w = stack.enter_context(wave.open(path, 'wb'))
                w.setnchannels(1)
                w.setsampwidth(2)
                w.setframerate(koala.sample_rate)
                return w
            out = open_wav(args.output_path)
            ref = open_wav(args.reference_output_path) if args.reference_output_path else None
            print('Listening... (press Ctrl+C to stop)')
            for frame in source:
                if args.realtime:
                    due = t_start + frames * n / koala.sample_rate
                    now = time.perf_counter()
                    if due > now:
                        time.sleep(due - now)
                t0 = time.perf_counter()
                enhanced = koala.process(frame)
                lat.append(time.perf_counter() - t0)
                out.writeframes(struct.pack('%dh' % n, *enhanced))
                if ref is not None:
                    ref.writeframes(struct.pack('%dh' % n, *[int(v) for v in frame]))
                frames += 1
    except KeyboardInterrupt:
        print('Stopping...')
    finally:
        koala.delete()
    if frames:
        a = np.array(lat) * 1e6
        print('%d frames (%.2f s of audio); process() latency p50 %.0f us, p99 %.0f us; delay %d samples' %
              (frames, frames * n / 16000.0, np.percentile(a, 50), np.percentile(a, 99), 256))
        print('Real time factor: %.4f' % (float(np.sum(lat)) / (frames * n / 16000.0)))
    return 0


if __name__ == '__main__':
    sys.exit(main())
