"""`main.py evaluation -g <ground truth> -c <computed>`: mean J, mean F and their mean over a results tree.

Same command line, same pairing rules and same three numbers as the reference's tool (src/evaluation.py:16-75):
  * the PNGs under the two folders are paired by sorted path;
  * inside a pair, the sorted distinct palette indices of the two images are paired POSITIONALLY (background included, so an
    object missing from one image shifts the pairing - kept, it is what the reference's numbers mean), the computed image is
    resized to the ground truth's size first, and a pair of images scores the mean (J, F) over its index pairs;
  * the result is the mean over image pairs.
The work is organised differently: image pairs are cut into contiguous chunks, every worker process scores its chunks with one
joint-histogram / boundary-stack pass per frame (metrics.frame_scores) and returns SUMS, and the parent only adds them up - no
per-image task objects, no score matrix of the whole dataset in memory.
"""
from concurrent.futures import ProcessPoolExecutor
from pathlib import Path

import click
import numpy as np

from .config import Config
from .metrics import frame_scores


def _index_maps(gt_path, seg_path):
    from PIL import Image
    gt = Image.open(gt_path).convert('P')
    seg = Image.open(seg_path).convert('P').resize(gt.size)
    return np.asarray(gt), np.asarray(seg)


def process_pair(gt, seg):
    """[J, F] of one (ground truth PNG, computed PNG) pair: the mean over the positionally paired palette indices."""
    gt_idx, seg_idx = _index_maps(gt, seg)
    pairs = list(zip(np.unique(gt_idx).tolist(), np.unique(seg_idx).tolist()))
    return frame_scores(gt_idx, seg_idx, pairs).mean(axis=0)


def _score_chunk(chunk):
    """Worker: (sum of [J, F] over the chunk's image pairs, number of pairs)."""
    total = np.zeros(2, dtype=np.float64)
    for gt, seg in chunk:
        total += process_pair(gt, seg)
    return total, len(chunk)


def _chunks(items, n):
    size = max(1, -(-len(items) // max(1, n)))
    return [items[i:i + size] for i in range(0, len(items), size)]


def evaluation_command_impl(ground_truth, computed_results, disable=False, processes=None):
    from tqdm import tqdm
    gt_files = sorted(Path(ground_truth).glob('**/*.png'))
    out_files = sorted(Path(computed_results).glob('**/*.png'))
    if len(gt_files) != len(out_files):
        raise AssertionError(f'{len(gt_files)} ground-truth images but {len(out_files)} computed ones')
    if not gt_files:
        raise AssertionError(f'no PNG files under {ground_truth}')
    workers = max(1, int(processes or Config.CPU_COUNT))
    todo = _chunks(list(zip(gt_files, out_files)), workers * 4)     # a few chunks per worker: progress + load balance
    total, count = np.zeros(2, dtype=np.float64), 0
    with tqdm(total=len(gt_files), disable=disable) as bar, ProcessPoolExecutor(max_workers=workers) as pool:
        for part, n in pool.map(_score_chunk, todo):
            total += part
            count += n
            bar.update(n)
    j_mean, f_mean = total / count
    return j_mean, f_mean, (j_mean + f_mean) / 2.0


@click.command(name='evaluation')
@click.option('--ground_truth', '-g', type=click.Path(file_okay=False, dir_okay=True), required=True,
              help='Path to ground truth dataset folder.')
@click.option('--computed_results', '-c', type=click.Path(file_okay=False, dir_okay=True), required=True,
              help='Path to computed results.')
def evaluation_command(ground_truth, computed_results):
    j_mean, f_mean, jf_mean = evaluation_command_impl(ground_truth, computed_results)
    click.echo(f'Evaluated: j_mean={j_mean}, f_mean={f_mean}, j&f_mean={jf_mean}.')
