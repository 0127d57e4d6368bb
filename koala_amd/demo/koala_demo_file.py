"""
File-to-file noise suppression with delay compensation -- the reference's file demo restated on koala_amd
(reference demo/python/koala_demo_file.py:21-137, loop :96-116; C twin demo/c/koala_demo_file.c:466-527), plus a
many-files mode that packs utterances into one batch handle.

    python -m koala_amd.demo.koala_demo_file --input_path noisy.wav --output_path clean.wav
    python -m koala_amd.demo.koala_demo_file --input_path a.wav b.wav c.wav --output_dir out/ --frames_per_call 32
"""
import argparse
import os
import struct
import time
import wave

import numpy as np

import koala_amd


def read_wav(path, sample_rate):
    with wave.open(path, 'rb') as f:
        if f.getframerate() != sample_rate:
            raise ValueError('Invalid sample rate of `%d`. Koala only accepts `%d`' % (f.getframerate(), sample_rate))
        if f.getnchannels() != 1:
            raise ValueError('This demo can only process single-channel WAV files')
        if f.getsampwidth() != 2:
            raise ValueError('This demo can only process WAV files with 16-bit PCM encoding')
        return np.frombuffer(f.readframes(f.getnframes()), dtype=np.int16).copy()


def write_wav(path, pcm, sample_rate):
    with wave.open(path, 'wb') as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sample_rate)
        f.writeframes(struct.pack('%dh' % len(pcm), *[int(x) for x in pcm]))


def enhance_single(koala, pcm):
    """One stream through Koala.process, frame by frame: zero-padded tail, output trimmed by delay_sample."""
    n, delay, length = koala.frame_length, koala.delay_sample, len(pcm)
    out = []
    start = 0
    while start < length + delay:
        end = start + n
        frame = np.zeros(n, np.int16)
        if start < length:
            chunk = pcm[start:min(end, length)]
            frame[:len(chunk)] = chunk
        y = koala.process(frame)
        if end > delay:
            if end > length + delay:
                y = y[:length + delay - start]
            if start < delay:
                y = y[delay - start:]
            out.extend(y)
        start = end
    return np.array(out, np.int16)


def enhance_batch(batch, signals, frames_per_call, asynchronous=False):
    """Many utterances at once: each is one stream of the batch, zero-padded to the longest; same trimming."""
    n, delay = batch.frame_length, batch.delay_sample
    longest = max(len(s) for s in signals)
    total_frames = -(-(longest + delay) // n)
    total_frames = -(-total_frames // frames_per_call) * frames_per_call
    x = np.zeros((batch.num_streams, total_frames * n), np.int16)
    for i, s in enumerate(signals):
        x[i, :len(s)] = s
    calls = list(range(0, total_frames, frames_per_call))
    if asynchronous:
        # three page-locked buffer pairs in rotation (pv_koala_batch_process_chunk_async): call n's copies run under its neighbours'
        # kernels; before pair n % 3 is reused, wait(2) guarantees that call n - 3 has completed and its output can be taken
        pairs = [(batch.alloc_host(frames_per_call), batch.alloc_host(frames_per_call)) for _ in range(3)]
        outs = []
        for i, c in enumerate(calls):
            a, b = pairs[i % 3]
            if i >= 3:
                batch.wait(2)
                outs.append(b.copy())
            a[:] = x[:, c * n:(c + frames_per_call) * n]
            batch.process_async(a, b)
        batch.wait(0)
        for i in range(max(0, len(calls) - 3), len(calls)):
            outs.append(pairs[i % 3][1].copy())
        y = np.concatenate(outs, axis=1)
    else:
        y = np.concatenate([batch.process(np.ascontiguousarray(x[:, c * n:(c + frames_per_call) * n])) for c in calls], axis=1)
    return [y[i, delay:delay + len(s)] for i, s in enumerate(signals)]


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--access_key', default='koala-amd', help='kept for compatibility; not validated')
    p.add_argument('--input_path', nargs='+', required=True)
    p.add_argument('--output_path', help='output WAV (single input)')
    p.add_argument('--output_dir', help='output directory (several inputs)')
    p.add_argument('--library_path')
    p.add_argument('--model_path')
    p.add_argument('--device', default='best')
    p.add_argument('--frames_per_call', type=int, default=32)
    p.add_argument('--precision', default='fp32', choices=['fp32', 'bf16'])
    p.add_argument('--asynchronous', action='store_true', help='many-files mode: asynchronous calls over three page-locked buffer pairs')
    p.add_argument('--show_inference_devices', action='store_true')
    args = p.parse_args()
    if args.show_inference_devices:
        print('\n'.join(koala_amd.available_devices(library_path=args.library_path)))
        return
    t0 = time.perf_counter()
    if len(args.input_path) == 1 and args.output_path:
        koala = koala_amd.create(args.access_key, model_path=args.model_path, device=args.device,
                                 library_path=args.library_path)
        try:
            pcm = read_wav(args.input_path[0], koala.sample_rate)
            t0 = time.perf_counter()
            out = enhance_single(koala, pcm)
            dt = time.perf_counter() - t0
            write_wav(args.output_path, out, koala.sample_rate)
            sr = koala.sample_rate
        finally:
            koala.delete()
        seconds = len(pcm) / sr
    else:
        if not args.output_dir:
            raise SystemExit('--output_dir is required for several inputs')
        os.makedirs(args.output_dir, exist_ok=True)
        batch = koala_amd.create_batch(args.access_key, len(args.input_path), args.frames_per_call, args.precision,
                                       model_path=args.model_path, device=args.device, library_path=args.library_path)
        try:
            signals = [read_wav(pth, batch.sample_rate) for pth in args.input_path]
            t0 = time.perf_counter()
            outs = enhance_batch(batch, signals, args.frames_per_call, args.asynchronous)
            dt = time.perf_counter() - t0
            for pth, o in zip(args.input_path, outs):
                write_wav(os.path.join(args.output_dir, os.path.basename(pth)), o, batch.sample_rate)
            sr = batch.sample_rate
        finally:
            batch.delete()
        seconds = sum(len(s) for s in signals) / sr
    # reference demo/c/koala_demo_file.c:526-527: real time factor = processing time / audio time
    print('%.2f seconds of audio have been written. Real time factor: %.5f' % (seconds, dt / seconds))


if __name__ == '__main__':
    main()
