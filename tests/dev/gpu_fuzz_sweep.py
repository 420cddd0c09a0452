import sys; sys.path.insert(0,'.')
import numpy as np
from avdsp_amd import progbuilder as pb, runtime as rt
from oracle import pyoracle as po
from tests.fuzz_programs import IN_BASE, N_IN, N_OUT, random_program, random_chain_case, stress_input
bad=0; n=0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    for fmt in (2,3,4,5,6):
        prog=random_program(seed,fmt)
        fs, block = [48000,48000,96000][seed%3], [1,64,300][seed%3]
        x=pb.lcg_input(300,N_IN,fmt in (5,6),seed=seed)
        o=po.OracleProgram(fmt,prog,fs=fs,random=seed,dither=24); r=rt.Runtime(fmt,prog,fs=fs,random=seed,dither=24)
        if r.rc<0: continue
        want=o.run_block(x,N_OUT,IN_BASE,0,scratch_len=48,block=block); got=r.run_block(x,N_OUT,IN_BASE,0,block=block)
        r.sync_state(); nn=int(prog[1])+int(prog[2]); n+=1
        if not (got.view(np.uint32)==want.view(np.uint32)).all() or not (r.buf[12:nn]==o.buf[12:nn]).all():
            bad+=1; print('GENERIC MISMATCH seed',seed,'fmt',fmt)
        r.release()
    rng,C,S,T,fmin,fmax,gain,fs,nf,dither=random_chain_case(seed+100000)
    for fmt in (2,4,6):
        taps=0 if fmt==2 else T
        if S==0 and taps==0: continue
        prog=pb.synth_program(2 if fmt==2 else 6,C,S,taps,fmin,fmax,gain)
        x=stress_input(rng,nf,C,fmt==6); block=int(rng.choice([7,64,nf]))
        o=po.OracleProgram(fmt,prog,fs=fs,dither=dither); r=rt.Runtime(fmt,prog,fs=fs,dither=dither)
        want=o.run_block(x,C,C,0,block=block); got=r.run_block(x,C,C,0,block=block); n+=1
        if not (got.view(np.uint32)==want.view(np.uint32)).all() or not (r.sync_state()==o.state).all():
            bad+=1; print('CHAIN MISMATCH seed',seed,'fmt',fmt,C,S,taps)
        r.release()
print('runs',n,'bad',bad)
