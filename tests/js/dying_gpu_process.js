// Stand-in for lib/emf/EmfGpuProcess.js in tests/test_node_host.py: rank 0 answers 'init' and then waits,
// rank 1 is killed by a signal (code === null in the parent's 'exit' event) -- what a GPU fault, a SIGSEGV
// in native code or the OOM killer look like to EmfLord.trainOnGpus.
'use strict';
process.on('message', (m) => {
  if (m.cmd == 'init') {
    if (m.rank == 1) process.kill(process.pid, process.env.YCNR_TEST_DIE_SIGNAL || 'SIGKILL');
    else process.send({ evt: 'ready' });
  } else if (m.cmd == 'destroy') {
    process.exit(0);
  }
});
setTimeout(() => process.exit(0), 60000);
