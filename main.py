#!/usr/bin/env python3
"""`python main.py inference ...` - the reference's command line (main.py:13-23) for the path this project replaces, and
`python main.py evaluation ...` (J / F of saved masks).  train / validation belong to the
reference's training stack and are out of scope."""
import importlib
import sys
from pathlib import Path

import click

sys.path.insert(0, str(Path(__file__).resolve().parent))
_inf = importlib.import_module('semi-supervised-vos_amd.inference')
_ev = importlib.import_module('semi-supervised-vos_amd.evaluation')


@click.group(name='cli')
def cli():
    pass


cli.add_command(_inf.inference_command)
cli.add_command(_ev.evaluation_command)


def _out_of_scope(name):
    @click.command(name=name, context_settings=dict(ignore_unknown_options=True, allow_extra_args=True))
    def cmd():
        raise click.ClickException(f"'{name}' is part of the reference's training stack and is not rebuilt "
                                   'here; this project replaces the `inference` hot path only')
    return cmd


for _n in ('train', 'validation'):
    cli.add_command(_out_of_scope(_n))

if __name__ == '__main__':
    cli()
