"""Command line of the MI355X build.

    python run.py train    NAME [-c CONFIG] [--max_epoch N] [--synthetic STEPS]
    python run.py generate NAME [INDEX] [-c CONFIG] [-s SRC_DIR] [-o OUT_DIR] [--skip_dummy]
    python run.py evaluate NAME INDEX [-c CONFIG] [-i ivfpq] [-t icassp] [--test_seq_len '1 3 5 9 11 19']

The command / argument / option surface equals the reference's run.py:13-162 (held to it by
tests/test_golden_cli.py; one declared extension: `train --synthetic`; `evaluate -i` keeps the reference's default
'ivfpq', which -- like every approximate faiss type -- is served by the exact search with a notice).
Configuration files are looked up as ./config/<CONFIG>.yaml relative to the working directory, like the
reference does.  All three commands run on the HIP library (include/nafp.h).

More than one GPU: start one process per GPU,
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 run.py generate NAME
`generate` then shards the rows over the ranks (no collective on the data path); `train` splits the global batch.
"""
import os
import sys

import click
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_config(config_fname):
    path = os.path.join('.', 'config', f'{config_fname}.yaml')
    if not os.path.exists(path):
        sys.exit(f'cli: ERROR! Configuration file ./config/{config_fname}.yaml is missing!!')
    print(f'cli: Configuration from ./config/{config_fname}.yaml')
    with open(path) as fh:
        return yaml.safe_load(fh)


def update_config(cfg, key1: str, key2: str, val):
    cfg.setdefault(key1, {})[key2] = val
    return cfg


def print_config(cfg):
    sys.stdout.write('\033[36m' + yaml.dump(cfg, indent=4, width=120, sort_keys=False) + '\033[0m\n')


def _init_distributed():
    """One process per GPU under torch.distributed.run; nothing to do for a single process."""
    if int(os.environ.get('WORLD_SIZE', '1')) <= 1:
        return
    import torch
    import torch.distributed as dist
    local = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local)
    dist.init_process_group('nccl', device_id=torch.device('cuda', local))


# ---- option tables: (flags, click keyword arguments) --------------------------------------------------
_CONFIG = (('--config', '-c'), dict(default='default', type=click.STRING, help='configuration name: ./config/<name>.yaml'))
_TRAIN = [
    _CONFIG,
    (('--max_epoch',), dict(default=None, type=click.INT, help='overrides TRAIN.MAX_EPOCH')),
    (('--synthetic',), dict(default=None, type=click.INT,
                            help='N steps per epoch on seeded noise anchors and replicas at 5 dB SNR instead of the '
                                 "configuration's training set")),
]
_GENERATE = [
    (('--config', '-c'), dict(default='default', required=False, type=click.STRING, help=_CONFIG[1]['help'])),
    (('--source', '-s'), dict(default=None, required=False, type=click.STRING,
                              help='fingerprint every 16-bit 8 kHz mono WAV below this directory instead of the test sets')),
    (('--output', '-o'), dict(default=None, required=False, type=click.STRING,
                              help='where <NAME>/<INDEX>/*.mm go; default DIR.OUTPUT_ROOT_DIR of the configuration')),
    (('--skip_dummy',), dict(default=False, is_flag=True, help='leave the dummy DB out of the default sources')),
]
_EVALUATE = [
    (('--config', '-c'), dict(default='default', required=False, type=click.STRING)),
    (('--index_type', '-i'), dict(default='ivfpq', type=click.STRING,
                                  help="'L2': exact search over the table resident in HBM (= faiss.IndexFlatL2).  The faiss "
                                       "types IVF, IVFPQ (the reference's default), IVFPQ-RR, IVFPQ-ONDISK, HNSW are accepted "
                                       "and served by the same exact search, with a notice.")),
    (('--test_seq_len',), dict(default='1 3 5 9 11 19', type=click.STRING,
                               help='query lengths in segments, space separated (1 3 5 9 11 19 = 1, 2, 3, 5, 6, 10 s)')),
    (('--test_ids', '-t'), dict(default='icassp', type=click.STRING,
                                help="'icassp' (the 2,000 ids of test_ids_icassp2021.npy), 'all', a .npy file of ids, or a "
                                     'number of random ids')),
    (('--nogpu',), dict(default=False, is_flag=True, help='accepted for compatibility; there is no CPU search here')),
]


def _with(options):
    def wrap(fn):
        for flags, kw in reversed(options):
            fn = click.option(*flags, **kw)(fn)
        return fn
    return wrap


@click.group()
def cli():
    """train -> generate -> evaluate on the MI355X library (`COMMAND --help` for the options)."""


@cli.command()
@click.argument('checkpoint_name', required=True)
@_with(_TRAIN)
def train(checkpoint_name, config, max_epoch, synthetic):
    """Contrastive training: device-side batches + augmentation, HIP forward / backward, NT-Xent or triplet loss,
    Adam / LAMB.  Under torch.distributed.run TR_BATCH_SZ is the global batch."""
    from neural_audio_fp_amd.model.trainer import synthetic_batches, trainer
    cfg = load_config(config)
    if max_epoch:
        update_config(cfg, 'TRAIN', 'MAX_EPOCH', max_epoch)
    print_config(cfg)
    _init_distributed()
    if synthetic is None:
        trainer(cfg, checkpoint_name)
    else:
        trainer(cfg, checkpoint_name, train_batches=synthetic_batches(cfg, synthetic), steps_per_epoch=synthetic)


@cli.command()
@click.argument('checkpoint_name', required=True)
@click.argument('checkpoint_index', required=False)
@_with(_GENERATE)
def generate(checkpoint_name, checkpoint_index, config, source, output, skip_dummy):
    """Fingerprints of the test sets (or of --source) from a checkpoint; without CHECKPOINT_INDEX the newest one."""
    from neural_audio_fp_amd.model.generate import generate_fingerprint
    cfg = load_config(config)
    _init_distributed()
    generate_fingerprint(cfg, checkpoint_name, checkpoint_index, source, output, skip_dummy)


@cli.command()
@click.argument('checkpoint_name', required=True)
@click.argument('checkpoint_index', required=True)
@_with(_EVALUATE)
def evaluate(checkpoint_name, checkpoint_index, config, index_type, test_seq_len, test_ids, nogpu):
    """Segment / sequence search over the generated {query, db, dummy_db}.mm and the hit-rate table."""
    from neural_audio_fp_amd.eval.eval_faiss import eval_faiss
    cfg = load_config(config)
    emb_dir = f"{cfg['DIR']['OUTPUT_ROOT_DIR']}{checkpoint_name}/{checkpoint_index}/"
    eval_faiss(emb_dir, index_type=index_type, test_seq_len=test_seq_len, test_ids=test_ids, nogpu=nogpu)


if __name__ == '__main__':
    cli()
