"""`main.py evaluation` of the reference (src/evaluation.py:16-75): mean J, mean F and their mean over all
(ground truth, computed) PNG pairs found under the two folders, paired by sorted order.

Kept as the reference does it, including its pairing rule inside a frame (process_pair, :16-37): the sorted distinct colours of
the two images are zipped POSITIONALLY - background included, and an object missing from one image shifts the pairing."""
from multiprocessing import Pool
from pathlib import Path

import click
import numpy as np

from .config import Config
from .metrics import evaluate_segmentation


def process_pair(gt, seg):
    from PIL import Image
    gt_img = Image.open(gt).convert('P')
    seg_img = Image.open(seg).convert('P')
    seg_img = seg_img.resize(gt_img.size)
    gt_img = np.asarray(gt_img)
    seg_img = np.asarray(seg_img)
    scores = []
    for gt_color, seg_color in zip(np.unique(gt_img), np.unique(seg_img)):
        scores.append(evaluate_segmentation(gt_img == gt_color, seg_img == seg_color))
    return np.array(scores).mean(axis=0)


@click.command(name='evaluation')
@click.option('--ground_truth', '-g', type=click.Path(file_okay=False, dir_okay=True), required=True,
              help='Path to ground truth dataset folder.')
@click.option('--computed_results', '-c', type=click.Path(file_okay=False, dir_okay=True), required=True,
              help='Path to computed results.')
def evaluation_command(ground_truth, computed_results):
    j_mean, f_mean, jf_mean = evaluation_command_impl(ground_truth, computed_results)
    click.echo(f'Evaluated: j_mean={j_mean}, f_mean={f_mean}, j&f_mean={jf_mean}.')


def evaluation_command_impl(ground_truth, computed_results, disable=False, processes=None):
    from tqdm import tqdm
    ground_truth = sorted(Path(ground_truth).glob('**/*.png'))
    computed = sorted(Path(computed_results).glob('**/*.png'))
    total = len(ground_truth)
    assert len(ground_truth) == len(computed), f'{total} ground-truth images but {len(computed)} computed ones'
    pbar = tqdm(total=total, disable=disable)
    with Pool(processes or Config.CPU_COUNT) as pool:
        res = [pool.apply_async(process_pair, args=(gt, seg), callback=lambda _: pbar.update(1))
               for gt, seg in zip(ground_truth, computed)]
        scores = np.array([p.get() for p in res])
    pbar.close()
    j_mean = scores[:, 0].mean()
    f_mean = scores[:, 1].mean()
    return j_mean, f_mean, np.array([j_mean, f_mean]).mean()
