"""Command line of the MI355X build: the reference's `run.py` surface.

    python run.py generate CHECKPOINT_NAME [CHECKPOINT_INDEX] [-c CONFIG] [-s SRC] [-o OUT] [--skip_dummy]
    python run.py train    CHECKPOINT_NAME [-c CONFIG] [--max_epoch N]
    python run.py evaluate CHECKPOINT_NAME CHECKPOINT_INDEX [-c CONFIG] [-i INDEX_TYPE] ...

Same commands, arguments, options and config resolution (./config/<name>.yaml) as
the reference's run.py:13-162.  `generate` and `train` run the HIP hot path (train on a
caller-supplied or synthetic batch source: the augmenting dataset is SURVEY.md section 8f);
`evaluate` runs the exact ('L2') search on the device; the approximate faiss index types say so.

Multi-GPU generate: launch one process per GPU, e.g.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        run.py generate NAME
Rows are then sharded across the ranks (no collective on the data path).
"""
import os
import sys

import click
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def load_config(config_fname):
    config_filepath = './config/' + config_fname + '.yaml'
    if os.path.exists(config_filepath):
        print(f'cli: Configuration from {config_filepath}')
    else:
        sys.exit(f'cli: ERROR! Configuration file {config_filepath} is missing!!')
    with open(config_filepath, 'r') as f:
        cfg = yaml.safe_load(f)
    return cfg


def update_config(cfg, key1: str, key2: str, val):
    cfg[key1][key2] = val
    return cfg


def print_config(cfg):
    print('\033[36m' + yaml.dump(cfg, indent=4, width=120, sort_keys=False) + '\033[0m')


def _init_distributed():
    """One process per GPU when launched by torch.distributed.run; no-op otherwise."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        import torch
        import torch.distributed as dist
        local = int(os.environ.get('LOCAL_RANK', '0'))
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))


@click.group()
def cli():
    """train-> generate-> evaluate.  `python run.py COMMAND --help` for details."""


@cli.command()
@click.argument('checkpoint_name', required=True)
@click.option('--config', '-c', default='default', type=click.STRING,
              help="Name of model configuration located in './config/.'")
@click.option('--max_epoch', default=None, type=click.INT, help='Max epoch.')
@click.option('--synthetic', default=None, type=click.INT,
              help='Train on N synthetic steps per epoch (anchors: seeded noise; replicas: anchors + noise at 5 dB '
                   'SNR) instead of the training set of the config.')
def train(checkpoint_name, config, max_epoch, synthetic):
    """Train a neural audio fingerprinter (HIP forward + backward, NT-Xent, Adam/LAMB).

    Multi-GPU: launch with torch.distributed.run; TR_BATCH_SZ is the GLOBAL batch, split evenly."""
    from neural_audio_fp_amd.model.trainer import trainer, synthetic_batches
    cfg = load_config(config)
    if max_epoch:
        update_config(cfg, 'TRAIN', 'MAX_EPOCH', max_epoch)
    print_config(cfg)
    _init_distributed()
    if synthetic is None:
        trainer(cfg, checkpoint_name)          # the reference's training set under cfg['DIR'] (device-side loader)
    else:
        trainer(cfg, checkpoint_name, train_batches=synthetic_batches(cfg, synthetic), steps_per_epoch=synthetic)


@cli.command()
@click.argument('checkpoint_name', required=True)
@click.argument('checkpoint_index', required=False)
@click.option('--config', '-c', default='default', required=False, type=click.STRING,
              help="Name of the model configuration file located in 'config/'. Default is 'default'")
@click.option('--source', '-s', default=None, type=click.STRING, required=False,
              help='Custom source root directory. The source must be 16-bit 8 Khz mono WAV.')
@click.option('--output', '-o', default=None, type=click.STRING, required=False,
              help='Root directory where the generated embeddings (uncompressed) will be stored. '
                   'Default is OUTPUT_ROOT_DIR/CHECKPOINT_NAME defined in config.')
@click.option('--skip_dummy', default=False, is_flag=True, help='Exclude dummy-DB from the default source.')
def generate(checkpoint_name, checkpoint_index, config, source, output, skip_dummy):
    """Generate fingerprints from a saved checkpoint.

    If CHECKPOINT_INDEX is not specified, the latest checkpoint is loaded.  The default
    sources are [TEST_DUMMY_DB] and [TEST_QUERY_DB] of the config file.
    """
    from neural_audio_fp_amd.model.generate import generate_fingerprint
    cfg = load_config(config)
    _init_distributed()
    generate_fingerprint(cfg, checkpoint_name, checkpoint_index, source, output, skip_dummy)


@cli.command()
@click.argument('checkpoint_name', required=True)
@click.argument('checkpoint_index', required=True)
@click.option('--config', '-c', default='default', required=False, type=click.STRING)
@click.option('--index_type', '-i', default='L2', type=click.STRING,
              help="'L2' = exact search over the HBM-resident table (the reference's faiss.IndexFlatL2). The "
                   "approximate faiss types {'IVF', 'IVFPQ', 'IVFPQ-RR', 'IVFPQ-ONDISK', 'HNSW'} are not built "
                   "(the reference's default is 'ivfpq').")
@click.option('--test_seq_len', default='1 3 5 9 11 19', type=click.STRING,
              help="Numbers of segments to test, separated by spaces. Default '1 3 5 9 11 19' = 1s, 2s, 3s, 5s, 6s, 10s.")
@click.option('--test_ids', '-t', default='icassp', type=click.STRING,
              help="One of {'all', 'icassp', 'path/file.npy', (int)}: all ids, the 2,000 ids of "
                   "eval/test_ids_icassp2021.npy, a 1-D array file, or N random ids.")
@click.option('--nogpu', default=False, is_flag=True, help='(reference flag) CPU-only search: not built here.')
def evaluate(checkpoint_name, checkpoint_index, config, index_type, test_seq_len, test_ids, nogpu):
    """Search and evaluation over the generated {query, db, dummy_db}.mm (run.py:140-161 of the reference)."""
    from neural_audio_fp_amd.eval.eval_faiss import eval_faiss
    cfg = load_config(config)
    emb_dir = cfg['DIR']['OUTPUT_ROOT_DIR'] + checkpoint_name + '/' + str(checkpoint_index) + '/'
    eval_faiss(emb_dir, index_type=index_type, test_seq_len=test_seq_len, test_ids=test_ids, nogpu=nogpu)


if __name__ == '__main__':
    cli()
