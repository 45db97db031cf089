gfalign filter -g testFiles/random3.gaf -n testFiles/random3.filter_nodelist.ls -o gaf
embedded
+++Alignment summary+++: 
# alignments: 3
Average read length: 18.67
Average aligned sequence: 18.67
Alignment orientation (+/-): 4(100.00%):0(0.00%)
Average path length: 18.67
Average alignment quality: 80.00
Average matches #: 18.67
Average block length: 18.67
Primary alignments: 0
Secondary alignments: 0
Supplementary alignments: 0
Terminal supplementary alignments: 0
